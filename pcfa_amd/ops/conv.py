"""Convolution-shaped operators of the encoders, update blocks and decoders: conv3x3 (Winograd F(2x2,3x3) / F(4x4,3x3)),
conv3x3_cat, dense_block, conv_s2 (+ fused 1x1 downsample), conv_fewin, conv3x3_fewout, sepconv5, instance norm + ReLU, add +
ReLU (models/raft/extractor.py, update.py; models/PWCNet/PWCNet.py:29-38,234-323; models/FlowNet/submodules.py)."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _ptr, _ptr_off, current_lane


_sepconv_packs = {}  # id(weight) -> (weakref, version, fwd_packed, bwd_packed)


FEWIN_SHAPES = {(2, 7), (1, 7), (2, 5), (2, 3)}  # (Cin, ksize) instances of pcfa_conv_fewin_fwd
_FEWIN_PACKED = os.environ.get("PCFA_FEWIN_PACKED", "1") != "0"   # A/B switch (tools/dev)
_fewin_packs = {}  # id(weight) -> (weakref, version, packed)


def _fewin_packed(weight):
    """pcfa_conv_fewin_pack of a frozen [N, Cin, k, k] weight (MFMA operand order), cached per tensor version."""
    key = id(weight)
    hit = _fewin_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        lib = _hip.load()
        N, Cin, k, _ = weight.shape
        w = weight.detach().contiguous()
        packed = torch.empty(int(lib.pcfa_conv_fewin_packed_floats(Cin, N, k)), device=w.device, dtype=torch.float32)
        _call("pcfa_conv_fewin_pack", _ptr(w), _ptr(packed), Cin, N, k)
        hit = (weakref.ref(weight, lambda _r, k_=key: _fewin_packs.pop(k_, None)), weight._version, packed)
        _fewin_packs[key] = hit
    return hit[2]


class _Conv1x1(torch.autograd.Function):
    """y[b] = W x[b] (+ bias) for a FROZEN 1x1 / stride-1 convolution on the package's own fp32-MFMA product
    (pcfa_gemm_f32): the feature / context encoders' output layer (models/raft/extractor.py:146,186) and the mask head's
    second layer (update.py:118-121).  Opt-in (Config.conv1x1 = "hip"): the library runs these three products 20-30 %
    faster, but a closure without any library kernel cannot meet a library workspace in another lane's graph
    (attack_PCFA.PairsInFlight) and does not depend on which solution the library picks for a shape."""

    @staticmethod
    def forward(ctx, x, w, b):
        _dev(x, w)
        if w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or x.dim() != 4 or x.shape[1] != w.shape[1]:
            raise ValueError("conv1x1: weight %s does not fit input %s" % (tuple(w.shape), tuple(x.shape)))
        if x.dtype != torch.float32 or w.dtype != torch.float32:
            raise ValueError("conv1x1: float32 only")
        x = x.contiguous()
        B, K, H, W = x.shape
        N, hw = w.shape[0], H * W
        w2 = w.reshape(N, K).contiguous()
        out = torch.empty((B, N, H, W), device=x.device, dtype=torch.float32)
        # C[b][n][p] = sum_k W[n][k] X[b][k][p]: A = W stored [M][K] (shared by the batch: stride 0), B = X[b] stored [K][N]
        _call("pcfa_gemm_f32", _ptr(w2), _ptr(x), _ptr(out), N, hw, K, K, hw, hw, 0, 1, B, 0, K * hw, N * hw, 1.0, 1,
              _ptr(None), ctypes.c_size_t(0))
        if b is not None:
            out.add_(b.view(1, N, 1, 1))
        ctx.save_for_backward(w2)
        ctx.dims = (B, K, N, H, W)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("conv1x1 is the frozen-weight path: no weight / bias gradient")
        (w2,) = ctx.saved_tensors
        B, K, N, H, W = ctx.dims
        hw = H * W
        g = g.contiguous()
        gx = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        # dX[b][k][p] = sum_n W[n][k] G[b][n][p]: A = W stored [K' = N][M' = K], B = G[b] stored [K' = N][N' = hw]
        _call("pcfa_gemm_f32", _ptr(w2), _ptr(g), _ptr(gx), K, hw, N, K, hw, hw, 1, 1, B, 0, N * hw, K * hw, 1.0, 1,
              _ptr(None), ctypes.c_size_t(0))
        return gx, None, None


def conv1x1(x, weight, bias=None):
    """conv2d(x, weight[N,K,1,1], bias), stride 1, frozen weights, without a library kernel (see _Conv1x1)."""
    return _Conv1x1.apply(x, weight, bias)


def conv_fewin(x, weight, bias=None, relu=False):
    """act(conv2d(x, weight, bias, stride=1, padding=k//2)) for a frozen k x k weight with <= 4 input channels and an
    input that needs no gradient (convf1 of the motion encoder on the detached flow): one streaming launch with bias
    and ReLU fused.  Forward only."""
    _dev(x, weight, bias)
    if x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad):
        raise RuntimeError("conv_fewin is forward-only: input and parameters must not require gradients")
    N, Cin, kh, kw = weight.shape
    if kh != kw or (Cin, kh) not in FEWIN_SHAPES or x.shape[1] != Cin:
        raise ValueError("conv_fewin: unsupported weight %s for input %s" % (tuple(weight.shape), tuple(x.shape)))
    x = x.contiguous()
    B, _, H, W = x.shape
    out = torch.empty((B, N, H, W), device=x.device, dtype=torch.float32)
    if _FEWIN_PACKED:
        _call("pcfa_conv_fewin_packed_fwd", _ptr(x), _ptr(_fewin_packed(weight)), _ptr(bias), _ptr(out), B, Cin, N, H, W,
              kh, int(bool(relu)))
    else:
        _call("pcfa_conv_fewin_fwd", _ptr(x), _ptr(weight.contiguous()), _ptr(bias), _ptr(out), B, Cin, N, H, W, kh,
              int(bool(relu)))
    return out


_s2_packs = {}  # id(weight) -> (weakref, version, packed)


def _s2_packed(weight):
    """pcfa_conv_s2_pack of a frozen [N, Cin, k, k] weight (MFMA operand order), cached per tensor version."""
    key = id(weight)
    hit = _s2_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        lib = _hip.load()
        N, Cin, k, _ = weight.shape
        w = weight.detach().contiguous()
        packed = torch.empty(int(lib.pcfa_conv_s2_packed_floats(Cin, N, k)), device=w.device, dtype=torch.float32)
        _call("pcfa_conv_s2_pack", _ptr(w), _ptr(packed), Cin, N, k)
        hit = (weakref.ref(weight, lambda _r, k_=key: _s2_packs.pop(k_, None)), weight._version, packed)
        _s2_packs[key] = hit
    return hit[2]


_s2_bwd_packs = {}


def _s2_bwd_packed(weight):
    key = id(weight)
    hit = _s2_bwd_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        lib = _hip.load()
        N, Cin, k, _ = weight.shape
        w = weight.detach().contiguous()
        packed = torch.empty(int(lib.pcfa_conv_s2_bwd_packed_floats(Cin, N, k)), device=w.device, dtype=torch.float32)
        _call("pcfa_conv_s2_bwd_pack", _ptr(w), _ptr(packed), Cin, N, k)
        hit = (weakref.ref(weight, lambda _r, k_=key: _s2_bwd_packs.pop(k_, None)), weight._version, packed)
        _s2_bwd_packs[key] = hit
    return hit[2]


def conv_s2_supported(x, weight):
    """True when conv_s2 covers conv2d(x, weight, stride=2, padding=k//2): the 3-channel 7x7 stem or any 3x3, W % 4 == 0."""
    N, Cin, kh, kw = weight.shape
    return bool(x.dim() == 4 and kh == kw and x.shape[1] == Cin and not weight.requires_grad and x.is_cuda
                and _hip.load().pcfa_conv_s2_supported(Cin, N, kh, x.shape[2], x.shape[3]))


class _ConvS2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope, grad_premasked=False, own_bwd=True):
        _dev(x, weight, bias)
        ctx.grad_premasked = bool(grad_premasked)   # the consumer applies this layer's activation backward (conv3x3)
        ctx.own_bwd = bool(own_bwd)                 # data gradient on pcfa_conv_s2_bwd where it applies (False: library)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        N, _, k, _ = weight.shape
        Ho, Wo = (H + 2 * (k // 2) - k) // 2 + 1, (W + 2 * (k // 2) - k) // 2 + 1
        out = torch.empty((B, N, Ho, Wo), device=x.device, dtype=torch.float32)
        _call("pcfa_conv_s2_fwd", _ptr(x), _ptr(_s2_packed(weight)), _ptr(bias), _ptr(out), B, Cin, N, H, W, k, act,
              float(slope))
        ctx.act, ctx.slope, ctx.xshape = act, float(slope), tuple(x.shape)
        ctx.save_for_backward(weight, out if (act and not ctx.grad_premasked) else None)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, out = ctx.saved_tensors
        g = g.contiguous()
        if ctx.act and not ctx.grad_premasked:
            gm = torch.empty_like(g)
            if ctx.act == 1:
                _call("pcfa_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), g.numel())
            else:
                _call("pcfa_leaky_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), ctx.slope, g.numel())
            g = gm
        N, Cin, k, _ = weight.shape
        B, _, H, W = ctx.xshape
        if ctx.own_bwd and _hip.load().pcfa_conv_s2_bwd_supported(Cin, N, k, H, W):
            gx = torch.empty(ctx.xshape, device=g.device, dtype=torch.float32)
            _call("pcfa_conv_s2_bwd", _ptr(g), _ptr(_s2_bwd_packed(weight)), _ptr(gx), B, Cin, N, H, W, k)
        else:   # the stem's gradient and ragged widths: library
            gx = torch.nn.grad.conv2d_input(ctx.xshape, weight, g, stride=2, padding=k // 2)
        return gx, None, None, None, None, None, None


_s2_ds_packs = {}   # (id(w), id(wd)) -> (weakref w, weakref wd, versions, fwd_packed, bwd_packed)


def _s2_ds_packed(weight, weight_d):
    key = (id(weight), id(weight_d))
    hit = _s2_ds_packs.get(key)
    ver = (weight._version, weight_d._version)
    if hit is None or hit[0]() is not weight or hit[1]() is not weight_d or hit[2] != ver:
        lib = _hip.load()
        N, Cin, _, _ = weight.shape
        w, wd = weight.detach().contiguous(), weight_d.detach().contiguous()
        pf = torch.empty(int(lib.pcfa_conv_s2_ds_packed_floats(Cin, N)), device=w.device, dtype=torch.float32)
        pb = torch.empty(int(lib.pcfa_conv_s2_ds_bwd_packed_floats(Cin, N)), device=w.device, dtype=torch.float32)
        _call("pcfa_conv_s2_ds_pack", _ptr(w), _ptr(wd), _ptr(pf), Cin, N)
        _call("pcfa_conv_s2_ds_bwd_pack", _ptr(w), _ptr(wd), _ptr(pb), Cin, N)
        drop = lambda _r, k_=key: _s2_ds_packs.pop(k_, None)
        hit = (weakref.ref(weight, drop), weakref.ref(weight_d, drop), ver, pf, pb)
        _s2_ds_packs[key] = hit
    return hit[3], hit[4]


def conv_s2_ds_supported(x, weight, weight_d):
    """True when conv_s2_ds covers the pair: a 3x3 and a 1x1 stride-2 convolution of the same input with equally many
    output channels, W % 8 == 0 (both directions on the HIP kernels)."""
    N, Cin, kh, kw = weight.shape
    if not (x.dim() == 4 and x.is_cuda and (kh, kw) == (3, 3) and tuple(weight_d.shape) == (N, Cin, 1, 1)
            and x.shape[1] == Cin and not weight.requires_grad and not weight_d.requires_grad):
        return False
    lib = _hip.load()
    return bool(lib.pcfa_conv_s2_supported(Cin, N, 3, x.shape[2], x.shape[3])
                and lib.pcfa_conv_s2_bwd_supported(Cin, N, 3, x.shape[2], x.shape[3]))


class _ConvS2DS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, weight_d, bias, bias_d, act):
        _dev(x, weight, weight_d, bias, bias_d)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        N = weight.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((B, N, Ho, Wo), device=x.device, dtype=torch.float32)
        out_d = torch.empty_like(out)
        pf, pb = _s2_ds_packed(weight, weight_d)
        _call("pcfa_conv_s2_ds_fwd", _ptr(x), _ptr(pf), _ptr(bias), _ptr(out), _ptr(bias_d), _ptr(out_d), B, Cin, N, H, W,
              act, 0.0)
        ctx.act, ctx.xshape, ctx.packed_bwd = act, tuple(x.shape), pb
        ctx.save_for_backward(out if act else None)
        return out, out_d

    @staticmethod
    def backward(ctx, g, gd):
        (out,) = ctx.saved_tensors
        B, Cin, H, W = ctx.xshape
        g = g.contiguous()
        gd = gd.contiguous()
        if ctx.act:
            gm = torch.empty_like(g)
            _call("pcfa_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), g.numel())
            g = gm
        gx = torch.empty(ctx.xshape, device=g.device, dtype=torch.float32)
        _call("pcfa_conv_s2_ds_bwd", _ptr(g), _ptr(gd), _ptr(ctx.packed_bwd), _ptr(gx), B, Cin, g.shape[1], H, W)
        return gx, None, None, None, None, None


def conv_s2_ds(x, weight, weight_d, bias=None, bias_d=None, relu=False):
    """(act(conv2d(x, weight, bias, stride=2, padding=1)), conv2d(x, weight_d, bias_d, stride=2)): conv1 and downsample[0]
    of a stride-2 residual block (extractor.py:23-58) in one launch per direction (pcfa_conv_s2_ds_fwd / _bwd)."""
    _dev(x, weight, weight_d, bias, bias_d)
    if not conv_s2_ds_supported(x, weight, weight_d):
        raise ValueError("conv_s2_ds: unsupported weights %s / %s for input %s"
                         % (tuple(weight.shape), tuple(weight_d.shape), tuple(x.shape)))
    if any(b is not None and b.requires_grad for b in (bias, bias_d)):
        raise RuntimeError("conv_s2_ds: frozen parameters only")
    return _ConvS2DS.apply(x, weight, weight_d, bias, bias_d, int(bool(relu)))


def conv_s2(x, weight, bias=None, relu=False, leaky_slope=None, grad_premasked=False, own_bwd=True):
    """act(conv2d(x, weight, bias, stride=2, padding=k//2)) for a frozen weight: the encoders' 7x7 stem and the 3x3
    first convolution of the down-sampling residual blocks on the fp32 matrix cores (pcfa_conv_s2_fwd)."""
    _dev(x, weight, bias)
    if weight.requires_grad or (bias is not None and bias.requires_grad):
        raise RuntimeError("conv_s2: frozen parameters only")
    if not conv_s2_supported(x, weight):
        raise ValueError("conv_s2: unsupported weight %s for input %s" % (tuple(weight.shape), tuple(x.shape)))
    act = 2 if leaky_slope is not None else int(bool(relu))
    if grad_premasked and not act:
        raise ValueError("conv_s2: grad_premasked needs an activation")
    return _ConvS2.apply(x, weight, bias, act, 0.0 if leaky_slope is None else leaky_slope, bool(grad_premasked),
                         bool(own_bwd))


class _Conv3x3FewOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, skip=False):
        _dev(x, weight, bias)
        ctx.set_materialize_grads(False)
        if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or not 1 <= weight.shape[0] <= 4:
            raise ValueError("conv3x3_fewout expects a [N<=4, K, 3, 3] weight, got %s" % (tuple(weight.shape),))
        x = x.contiguous()
        B, K, H, W = x.shape
        N = weight.shape[0]
        if weight.shape[1] != K:
            raise ValueError("conv3x3_fewout: input %s does not match weight %s" % (tuple(x.shape), tuple(weight.shape)))
        w = weight.detach().contiguous()
        out = torch.empty((B, N, H, W), device=x.device, dtype=torch.float32)
        nws = int(_hip.load().pcfa_conv3x3_fewout_workspace_bytes(B, K, N, H, W))
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32) if nws else None
        _call("pcfa_conv3x3_fewout_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), B, K, N, H, W)
        ctx.save_for_backward(w)
        ctx.dims = (B, K, N, H, W)
        if skip:
            return out, x.view_as(x)   # the alias feeds x's other consumer: its gradient is summed in the kernel below
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, g_skip=None):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("conv3x3_fewout is the frozen-weight path: no weight / bias gradient")
        if g is None:
            return g_skip, None, None, None
        (w,) = ctx.saved_tensors
        B, K, N, H, W = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_conv3x3_fewout_bwd", _ptr(g), _ptr(w), _ptr(None if g_skip is None else g_skip.contiguous()), _ptr(gx),
              B, K, N, H, W)
        return gx, None, None, None


def conv3x3_fewout(x, weight, bias=None, skip=False):
    """conv2d(x, weight, bias, stride=1, padding=1) for a frozen 3x3 weight with at most 4 output channels (the
    flow-prediction layers): a streaming kernel instead of a padded matrix-core tile.  skip=True returns (result, x_alias):
    hand x_alias to x's OTHER consumer (PWC-Net: upfeat beside predict_flow) and its gradient is added inside this layer's
    data-gradient kernel instead of by an autograd add over the whole tensor (as ops.conv3x3(skip=True))."""
    return _Conv3x3FewOut.apply(x, weight, bias, skip)


class _InstNormRelu(torch.autograd.Function):
    """relu?(F.instance_norm(x, eps=eps)) on pcfa_instnorm_fwd/bwd (two streaming launches per direction)."""

    @staticmethod
    def forward(ctx, x, eps, relu):
        _dev(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        planes, plane = B * C, H * W
        lib = _hip.load()
        ws = torch.empty((int(lib.pcfa_instnorm_workspace_bytes(planes, plane)) + 3) // 4, device=x.device,
                         dtype=torch.float32)
        y = torch.empty_like(x)
        stats = torch.empty((planes, 2), device=x.device, dtype=torch.float32)
        _call("pcfa_instnorm_fwd", _ptr(x), _ptr(y), _ptr(stats), _ptr(ws), planes, plane, float(eps), int(bool(relu)))
        ctx.save_for_backward(x, stats)
        ctx.dims = (planes, plane, int(bool(relu)))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, stats = ctx.saved_tensors
        planes, plane, relu = ctx.dims
        g = g.contiguous()
        lib = _hip.load()
        ws = torch.empty((int(lib.pcfa_instnorm_workspace_bytes(planes, plane)) + 3) // 4, device=x.device,
                         dtype=torch.float32)
        gx = torch.empty_like(x)
        _call("pcfa_instnorm_bwd", _ptr(x), _ptr(stats), _ptr(g), _ptr(gx), _ptr(ws), planes, plane, relu)
        return gx, None, None


def instance_norm_relu(x, eps=1e-5, relu=False):
    """relu?(InstanceNorm2d(affine=False, track_running_stats=False)(x)) -- models/raft/extractor.py:23-58."""
    return _InstNormRelu.apply(x, eps, relu)


class _AddRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, b_is_relu):
        _dev(a, b)
        if a.shape != b.shape:
            raise ValueError("add_relu: shapes differ: %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _call("pcfa_add_relu_fwd", _ptr(a), _ptr(b), _ptr(out), a.numel())
        ctx.b_is_relu = bool(b_is_relu)
        ctx.save_for_backward(out, *((b,) if ctx.b_is_relu else ()))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        out = ctx.saved_tensors[0]
        g = g.contiguous()
        gm = torch.empty_like(g)
        if ctx.b_is_relu:   # b = relu(.) of a layer that left its mask to us: its gradient is masked in the same pass
            gb = torch.empty_like(g)
            _call("pcfa_relu_bwd2", _ptr(out), _ptr(ctx.saved_tensors[1]), _ptr(g), _ptr(gm), _ptr(gb), g.numel())
            return gm, gb, None
        _call("pcfa_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), g.numel())
        return gm, gm, None


def add_relu(a, b, b_is_relu=False):
    """relu(a + b): the output of ResidualBlock.forward (models/raft/extractor.py:50-58).  b_is_relu: b is the ReLU output
    of a layer run with grad_premasked=True and has no other consumer -- the gradient returned for b is already
    multiplied by [b > 0] (one pass produces both gradients)."""
    return _AddRelu.apply(a, b, b_is_relu)


def _sepconv5_packed(weight):
    """pcfa_sepconv5_pack_weights of a frozen (1,5)/(5,1) Conv2d weight, cached per tensor version."""
    key = id(weight)
    hit = _sepconv_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        cout, cin = weight.shape[:2]
        w = weight.detach().contiguous()
        lib = _hip.load()   # direct order + Winograd-domain weights (csrc/sepconv5_wino.hip)
        fwd = torch.empty(int(lib.pcfa_sepconv5_packed_floats(cout, cin)), device=w.device, dtype=torch.float32)
        bwd = torch.empty(int(lib.pcfa_sepconv5_packed_floats(cin, cout)), device=w.device, dtype=torch.float32)
        _call("pcfa_sepconv5_pack_weights", _ptr(w), _ptr(fwd), _ptr(bwd), cout, cin)
        hit = (weakref.ref(weight, lambda _r, k=key: _sepconv_packs.pop(k, None)), weight._version, fwd, bwd)
        _sepconv_packs[key] = hit
    return hit[2], hit[3]


class _SepConv5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        _dev(a, b, weight)
        if weight.dim() != 4 or tuple(weight.shape[2:]) not in ((1, 5), (5, 1)) or weight.dtype != torch.float32:
            raise ValueError("sepconv5 expects a float32 (1,5) or (5,1) Conv2d weight, got %s" % (tuple(weight.shape),))
        vertical = int(weight.shape[2] == 5)
        a = a.contiguous()
        b = None if b is None else b.contiguous()
        B, Ca, H, W = a.shape
        Cb = 0 if b is None else b.shape[1]
        cout = weight.shape[0]
        if weight.shape[1] != Ca + Cb or (b is not None and (b.shape[0], b.shape[2], b.shape[3]) != (B, H, W)):
            raise ValueError("sepconv5: operands %s / %s do not match weight %s"
                             % (tuple(a.shape), None if b is None else tuple(b.shape), tuple(weight.shape)))
        fwd, bwd = _sepconv5_packed(weight)
        out = torch.empty((B, cout, H, W), device=a.device, dtype=torch.float32)
        _call("pcfa_sepconv5_fwd", _ptr(a), Ca, _ptr(b), Cb, _ptr(fwd), _ptr(out), B, cout, H, W, vertical)
        ctx.bwd, ctx.dims = bwd, (B, Ca, Cb, cout, H, W, vertical)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.needs_input_grad[2]:
            raise RuntimeError("sepconv5 is the frozen-weight path: no weight gradient (use the module's "
                               "reference forward when training)")
        B, Ca, Cb, cout, H, W, vertical = ctx.dims
        g = grad_out.contiguous()
        gin = torch.empty((B, Ca + Cb, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_sepconv5_fwd", _ptr(g), cout, None, 0, _ptr(ctx.bwd), _ptr(gin), B, Ca + Cb, H, W, vertical)
        return gin[:, :Ca], (gin[:, Ca:] if Cb else None), None


_conv3_packs = {}  # id(weight) -> (weakref, version, fwd_packed, bwd_packed)


def _conv3x3_packed(weight):
    """pcfa_conv3x3_pack_weights of a frozen 3x3 Conv2d weight (Winograd-transformed, both directions), cached per
    tensor version."""
    key = id(weight)
    hit = _conv3_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        cout, cin = weight.shape[:2]
        lib = _hip.load()
        w = weight.detach().contiguous()
        fwd = torch.empty(int(lib.pcfa_conv3x3_packed_floats(cin, cout)), device=w.device, dtype=torch.float32)
        bwd = torch.empty(int(lib.pcfa_conv3x3_packed_floats(cout, cin)), device=w.device, dtype=torch.float32)
        _call("pcfa_conv3x3_pack_weights", _ptr(w), _ptr(fwd), _ptr(bwd), cout, cin)
        hit = (weakref.ref(weight, lambda _r, k=key: _conv3_packs.pop(k, None)), weight._version, fwd, bwd)
        _conv3_packs[key] = hit
    return hit[2], hit[3]


_CONV_WS = {}   # device index -> scratch of pcfa_conv3x3_run (split-K partial outputs of the F(4x4,3x3) path)
_CONV_WS_RETIRED = []   # superseded (smaller) scratch buffers: kept alive for the graphs that captured their address


_SIDE_STREAMS = {}          # (device index, lane) -> the lane's second stream
_SIDE_HANDLES = set()       # their HIP handles


def side_stream(device):
    """The second stream on which nets/raft.py runs the context encoder beside the feature encoder: one per (device, lane),
    created on first use and bound to its lane (core.bind_stream) -- autograd runs the encoder's backward nodes on a thread
    of its own, where only the stream says whose scratch buffers a launch may use.  (r05: one stream per DEVICE made the
    context encoder's backward of every lane take lane 0's K-slice scratch; two pairs in flight at a small map size raced
    on it -- tests/test_gpu_parity.py::test_pairs_in_flight_bit_identical_to_solo caught it in the full-suite order.)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    k = current_lane()
    s = _SIDE_STREAMS.get((idx, k))
    if s is None:
        s = _SIDE_STREAMS[(idx, k)] = core.own_stream(device, k)
        _SIDE_HANDLES.add(s.cuda_stream)
    return s


def _conv_workspace(device, nbytes):
    """One scratch buffer per (device, main | side stream, lane), grown on demand OUTSIDE graph captures (every capture in this package
    follows eager warm-up calls of the same shapes); convolutions are stream-ordered per stream, and two streams (the
    encoders running side by side, nets/raft.py) never share a buffer."""
    dev_idx = device.index if device.index is not None else torch.cuda.current_device()
    idx = (dev_idx, torch.cuda.current_stream(device).cuda_stream in _SIDE_HANDLES, current_lane())
    ws = _CONV_WS.get(idx)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("conv3x3 workspace would have to grow inside a graph capture (no eager warm-up of this "
                               "shape ran before it)")
        if ws is not None:
            # never free a scratch buffer a captured hipGraph may have baked in (pcfa_conv3x3_run's split-K partials):
            # graphs are kept across pairs (attack_PCFA._PairGraphs), and a replay after the buffer grew for another
            # shape would write into memory the allocator has handed to someone else
            _CONV_WS_RETIRED.append(ws)
        ws = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _CONV_WS[idx] = ws
    return ws


def _conv3x3_run(device, x_ptr, packed, bias_ptr, mask_ptr, addend_ptr, out_ptr, B, K, N, H, W, act=0, slope=0.,
                 mask_channels=0):
    """out = act(bias + conv3x3(x)) [masked] [+ addend] through pcfa_conv3x3_run (the library picks F(4x4,3x3) or
    F(2x2,3x3) per shape); pointers are raw device addresses (or None).  With a mask (act = 0) `slope` is the factor where
    the mask is not positive; mask_channels > 0: only that channel prefix, after the addend (include/pcfa_hip.h)."""
    nws = int(_hip.load().pcfa_conv3x3_workspace_bytes(B, K, N, H, W))
    ws = _conv_workspace(device, nws) if nws else None
    _call("pcfa_conv3x3_run", x_ptr, _ptr(packed), bias_ptr, mask_ptr, addend_ptr, out_ptr, B, K, N, H, W, int(act),
          float(slope), int(mask_channels), _ptr(ws), nws)


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu, slope=None, skip=False, flags=0, input_slope=0.):
        _dev(x, weight, bias)
        ctx.skip = bool(skip)
        ctx.grad_premasked, ctx.mask_input_grad = bool(flags & 1), bool(flags & 2)
        ctx.input_slope = float(input_slope)   # slope of the (Leaky)ReLU that produced x (mask_input_grad; 0 = ReLU)
        ctx.set_materialize_grads(False)
        if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or weight.dtype != torch.float32:
            raise ValueError("conv3x3 expects a float32 3x3 Conv2d weight, got %s" % (tuple(weight.shape),))
        x = x.contiguous()
        B, K, H, W = x.shape
        N = weight.shape[0]
        if weight.shape[1] != K:
            raise ValueError("conv3x3: input %s does not match weight %s" % (tuple(x.shape), tuple(weight.shape)))
        fwd, bwd = _conv3x3_packed(weight)
        out = torch.empty((B, N, H, W), device=x.device, dtype=torch.float32)
        act = 2 if slope is not None else int(bool(relu))
        _conv3x3_run(x.device, _ptr(x), fwd, _ptr(bias), None, None, _ptr(out), B, K, N, H, W, act, float(slope or 0.))
        ctx.bwd, ctx.dims, ctx.act, ctx.slope = bwd, (B, K, N, H, W), act, float(slope or 0.)
        if ctx.grad_premasked and not act:
            raise ValueError("conv3x3: grad_premasked needs an activation (relu=True or leaky_slope)")
        ctx.save_for_backward(*(([out] if act and not ctx.grad_premasked else []) + ([x] if ctx.mask_input_grad else [])))
        if ctx.skip:
            return out, x.view_as(x)   # the alias feeds the residual path: its gradient is summed in the epilogue below
        return out

    @staticmethod
    def backward(ctx, g, g_skip=None):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("conv3x3 is the frozen-weight path: no weight / bias gradient")
        B, K, N, H, W = ctx.dims
        if g is None:
            return (None if g_skip is None else g_skip), None, None, None, None, None, None, None
        g = g.contiguous()
        if ctx.act and not ctx.grad_premasked:
            out = ctx.saved_tensors[0]
            gm = torch.empty_like(g)
            if ctx.act == 1:
                _call("pcfa_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), g.numel())
            else:
                _call("pcfa_leaky_relu_bwd", _ptr(out), _ptr(g), _ptr(gm), ctx.slope, g.numel())
            g = gm
        gin = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        if g_skip is not None or ctx.mask_input_grad:
            xin = ctx.saved_tensors[-1] if ctx.mask_input_grad else None   # = a (Leaky)ReLU output: [xin > 0] is its mask
            _conv3x3_run(g.device, _ptr(g), ctx.bwd, None, _ptr(xin),
                         _ptr(None if g_skip is None else g_skip.contiguous()), _ptr(gin), B, N, K, H, W,
                         slope=ctx.input_slope if ctx.mask_input_grad else 0.)
        else:
            _conv3x3_run(g.device, _ptr(g), ctx.bwd, None, None, None, _ptr(gin), B, N, K, H, W)
        return gin, None, None, None, None, None, None, None


_PAIR_LAUNCH = os.environ.get("PCFA_CONV3X3_PAIR", "1") != "0"   # A/B switch (tools/dev)


class _Conv3x3Cat(torch.autograd.Function):
    """cat([relu(conv3x3(x_i, w_i, b_i)) for i] + tails, dim=1) in one pre-allocated buffer (batch size 1): the
    convolutions write their channel blocks in place, trailing tensors (e.g. the flow of the motion encoder,
    models/raft/update.py:91-101) are copied behind them -- no torch.cat pass over the convolution outputs."""

    @staticmethod
    def forward(ctx, n_conv, flags, *args):
        xs, ws, bs = args[0:3 * n_conv:3], args[1:3 * n_conv:3], args[2:3 * n_conv:3]
        tails = args[3 * n_conv:]
        ctx.grad_premasked, ctx.mask_input_grads = bool(flags & 1), bool(flags & 2)
        _dev(*xs, *ws, *tails)
        xs = [x.contiguous() for x in xs]
        B, _, H, W = xs[0].shape
        if B != 1:
            raise ValueError("conv3x3_cat: batch size 1 only")
        plane = H * W
        widths = [w.shape[0] for w in ws] + [t.shape[1] for t in tails]
        buf = torch.empty((1, sum(widths), H, W), device=xs[0].device, dtype=torch.float32)
        off, packs, fwds, offs = 0, [], [], []
        for x, w, b in zip(xs, ws, bs):
            if tuple(x.shape) != (1, w.shape[1], H, W) or tuple(w.shape[2:]) != (3, 3):
                raise ValueError("conv3x3_cat: input %s does not fit weight %s" % (tuple(x.shape), tuple(w.shape)))
            fwd, bwd = _conv3x3_packed(w)
            fwds.append(fwd)
            offs.append(off)
            packs.append((bwd, w.shape[1], w.shape[0], off))
            off += w.shape[0]
        lib = _hip.load()
        f23 = all(lib.pcfa_conv3x3_algo(1, w.shape[1], w.shape[0], H, W) == 23 for w in ws)
        if n_conv == 2 and (ws[0].shape[1] % 8 == 0) == (ws[1].shape[1] % 8 == 0) and _PAIR_LAUNCH and f23:
            # two independent convolutions, one launch: the smaller one's workgroups fill the larger one's last round
            i, j = (0, 1) if ws[0].shape[0] * ws[0].shape[1] >= ws[1].shape[0] * ws[1].shape[1] else (1, 0)
            _call("pcfa_conv3x3_act_fwd_pair", _ptr(xs[i]), _ptr(fwds[i]), _ptr(bs[i]), _ptr_off(buf, offs[i] * plane),
                  ws[i].shape[1], ws[i].shape[0], _ptr(xs[j]), _ptr(fwds[j]), _ptr(bs[j]),
                  _ptr_off(buf, offs[j] * plane), ws[j].shape[1], ws[j].shape[0], H, W, 1, 0.)
        else:
            for x, w, b, fwd, o_ in zip(xs, ws, bs, fwds, offs):
                _conv3x3_run(x.device, _ptr(x), fwd, _ptr(b), None, None, _ptr_off(buf, o_ * plane), 1, w.shape[1],
                             w.shape[0], H, W, 1, 0.)
        for t in tails:
            buf[:, off:off + t.shape[1]].copy_(t)
            off += t.shape[1]
        ctx.packs, ctx.dims, ctx.n_conv, ctx.tail_widths = packs, (H, W), n_conv, [t.shape[1] for t in tails]
        ctx.save_for_backward(buf, *(xs if ctx.mask_input_grads else ()))
        return buf

    @staticmethod
    def backward(ctx, g):
        buf = ctx.saved_tensors[0]
        H, W = ctx.dims
        plane = H * W
        g = g.contiguous()
        grads = [None, None]
        for i, (bwd, k, n, off) in enumerate(ctx.packs):
            if ctx.needs_input_grad[3 + 3 * i] or ctx.needs_input_grad[4 + 3 * i]:
                raise RuntimeError("conv3x3_cat is the frozen-weight path: no weight / bias gradient")
            gx = None
            if ctx.needs_input_grad[2 + 3 * i]:
                if ctx.grad_premasked:      # the consumer already applied this layer's ReLU mask to its gradient
                    gm = g[:, off:off + n]  # a channel block of a batch-1 NCHW tensor: contiguous
                else:
                    gm = torch.empty((1, n, H, W), device=g.device, dtype=torch.float32)
                    _call("pcfa_relu_bwd", _ptr_off(buf, off * plane), _ptr_off(g, off * plane), _ptr(gm), n * plane)
                gx = torch.empty((1, k, H, W), device=g.device, dtype=torch.float32)
                if ctx.mask_input_grads:    # x_i is a ReLU output whose producer left its mask to this epilogue
                    _conv3x3_run(g.device, _ptr(gm), bwd, None, _ptr(ctx.saved_tensors[1 + i]), None, _ptr(gx), 1, n, k,
                                 H, W)
                else:
                    _conv3x3_run(g.device, _ptr(gm), bwd, None, None, None, _ptr(gx), 1, n, k, H, W)
            grads += [gx, None, None]
        off = sum(p[2] for p in ctx.packs)
        for j, tw in enumerate(ctx.tail_widths):
            grads.append(g[:, off:off + tw] if ctx.needs_input_grad[2 + 3 * ctx.n_conv + j] else None)
            off += tw
        return tuple(grads)


def conv3x3_cat(convs, tails=(), grad_premasked=False, mask_input_grads=False):
    """convs = [(x, weight, bias), ...] (frozen 3x3 / stride 1 / pad 1, ReLU), tails = tensors appended unchanged.
    Deferred ReLU masks (each saves one elementwise launch per layer and backward; the CALLER guarantees the contract):
    grad_premasked   -- every consumer of the result multiplies the gradient of the convolution channels by
                        [result > 0] itself (conv3x3_cat(mask_input_grads=True), gru_step(rest_relu_channels=...)), so
                        the backward here skips its ReLU pass;
    mask_input_grads -- every x_i is a ReLU output produced with grad_premasked=True: its mask [x_i > 0] is applied in
                        the epilogue of the data-gradient kernel."""
    flat = []
    for x, w, b in convs:
        flat += [x, w, b]
    return _Conv3x3Cat.apply(len(convs), int(bool(grad_premasked)) | 2 * int(bool(mask_input_grads)), *flat, *tails)


class _DenseBlock(torch.autograd.Function):
    """x_{i+1} = cat(leaky_relu(conv3x3_i(x_i)), x_i) for i = 0..n-1 (PWC-Net's DenseNet decoders, PWCNet.py:234-323)
    written into ONE pre-allocated buffer: every convolution reads the channel suffix it needs in place and writes
    its output in front of it, so no torch.cat copies the growing tensor (5 copies of up to 69 MB per level).
    Batch size 1 only (a channel suffix of an NCHW tensor is contiguous only then)."""

    @staticmethod
    def forward(ctx, slope, fused_masks, nparts, *rest):
        parts, wb = rest[:nparts], rest[nparts:]
        weights, biases = wb[0::2], wb[1::2]
        ctx.fused_masks = bool(fused_masks)
        _dev(*parts, *weights)
        x0 = parts[0]
        B, _, H, W = x0.shape
        if B != 1:
            raise ValueError("dense_block: batch size 1 only")
        if any(p.shape[0] != 1 or tuple(p.shape[2:]) != (H, W) or p.dtype != torch.float32 for p in parts):
            raise ValueError("dense_block: input parts %s do not share batch 1 / size / fp32" % ([tuple(p.shape) for p in parts],))
        ctx.part_widths = [p.shape[1] for p in parts]
        K0 = sum(ctx.part_widths)
        widths = [w.shape[0] for w in weights]
        total = K0 + sum(widths)
        buf = torch.empty((1, total, H, W), device=x0.device, dtype=torch.float32)
        plane = H * W
        start = total - K0
        if nparts == 1:
            buf[:, start:].copy_(x0)
        else:   # the caller's torch.cat((corr, f1, up_flow, up_feat), 1) lands in the block buffer directly
            torch.cat(parts, 1, out=buf[:, start:])
        packs = []
        k = K0
        for w, b, n in zip(weights, biases, widths):
            if tuple(w.shape[1:]) != (k, 3, 3):
                raise ValueError("dense_block: weight %s does not fit %d input channels" % (tuple(w.shape), k))
            fwd, bwd = _conv3x3_packed(w)
            _conv3x3_run(x0.device, _ptr_off(buf, start * plane), fwd, _ptr(b), None, None,
                         _ptr_off(buf, (start - n) * plane), 1, k, n, H, W, 2, float(slope))
            packs.append((bwd, k, n, start))
            start -= n
            k += n
        ctx.packs, ctx.dims, ctx.slope = packs, (total, K0, H, W), float(slope)
        ctx.save_for_backward(buf)
        return buf

    @staticmethod
    def backward(ctx, g):
        if any(ctx.needs_input_grad[3 + len(ctx.part_widths):]):
            raise RuntimeError("dense_block is the frozen-weight path: no weight / bias gradient")
        (buf,) = ctx.saved_tensors
        total, K0, H, W = ctx.dims
        plane = H * W
        # running gradient of the buffer: every layer adds its input gradient to a suffix.  `g` itself must not be written
        # (autograd may share it): the TOP layer reads its addend from g and writes the sum into gb -- its suffix is
        # everything any later step reads, so g is never copied (r03 cloned it: up to 69 MB per level)
        g = g.contiguous()
        gb = torch.empty_like(g)
        npk = len(ctx.packs)
        for i in range(npk - 1, -1, -1):
            bwd, k, n, start = ctx.packs[i]
            if i == npk - 1 or not ctx.fused_masks:
                # LeakyReLU backward of this layer's output (the top layer's gradient arrives from outside only)
                gm = torch.empty((1, n, H, W), device=g.device, dtype=torch.float32)
                _call("pcfa_leaky_relu_bwd", _ptr_off(buf, (start - n) * plane),
                      _ptr_off(g if i == npk - 1 else gb, (start - n) * plane), _ptr(gm), ctx.slope, n * plane)
                gm_ptr = _ptr(gm)
            else:   # already multiplied by the layer above (below): its slot of the running gradient IS the masked gradient
                gm_ptr = _ptr_off(gb, (start - n) * plane)
            # the layer's input gradient is added to the running gradient in the convolution's epilogue, in place (every
            # output element reads its own addend): no separate gradient tensor, no add launch.  The first channels of
            # the suffix are the output of layer i - 1, and this is the last contribution to their gradient: its
            # LeakyReLU backward rides in the same epilogue (mask = that layer's output in the block buffer, applied
            # after the addend) -- 4 elementwise launches less per block.
            dst = _ptr_off(gb, start * plane)
            src = _ptr_off(g, start * plane) if i == npk - 1 else dst
            if i > 0 and ctx.fused_masks:
                _conv3x3_run(g.device, gm_ptr, bwd, None, _ptr_off(buf, start * plane), src, dst, 1, n, k, H, W,
                             slope=ctx.slope, mask_channels=ctx.packs[i - 1][2])
            else:
                _conv3x3_run(g.device, gm_ptr, bwd, None, None, src, dst, 1, n, k, H, W)
        grads, c0 = [], total - K0
        for wdt, need in zip(ctx.part_widths, ctx.needs_input_grad[3:]):   # channel ranges of one image: contiguous views
            grads.append(gb[:, c0:c0 + wdt] if need else None)
            c0 += wdt
        return (None, None, None) + tuple(grads) + (None,) * (2 * len(ctx.packs))


def dense_block(x, layers, slope=0.1, fused_masks=True):
    """layers = [(weight, bias), ...] of frozen 3x3 convolutions; returns cat(y_n-1, ..., y_0, x) along channels.
    x: one tensor, or a tuple of tensors standing for their channel concatenation (PWCNet.py:265: cat((corr, c1, up_flow,
    up_feat), 1)) -- the parts are written into the block's buffer directly and get their gradients as views of it.
    fused_masks (Config.dense_block_fused_masks): the LeakyReLU backward of layers 0..n-2 rides in the data-gradient
    epilogue of the layer above; False = one pcfa_leaky_relu_bwd launch per layer."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    parts = tuple(x) if isinstance(x, (tuple, list)) else (x,)
    return _DenseBlock.apply(slope, fused_masks, len(parts), *[p.contiguous() for p in parts], *flat)


def conv3x3(x, weight, bias=None, relu=False, leaky_slope=None, skip=False, grad_premasked=False, mask_input_grad=False,
            input_slope=0.):
    """act(conv2d(x, weight, bias, stride=1, padding=1)) for a frozen 3x3 weight: Winograd F(2x2,3x3) on the fp32
    matrix cores with bias and ReLU (or LeakyReLU(leaky_slope)) fused into the epilogue; the data gradient runs the
    same kernel.  skip=True returns (result, x_alias): use x_alias for a residual connection around the convolution --
    the gradient arriving on it is added in the data-gradient kernel's epilogue instead of by an autograd `add`.
    grad_premasked / mask_input_grad: the deferred-ReLU contract of conv3x3_cat (the consumer of this layer's output
    applies [output > 0] to the gradient / this layer applies [x > 0] to the gradient it returns for a ReLU-output x);
    with LeakyReLU layers the factor where the output is not positive is the producer's slope (input_slope)."""
    return _Conv3x3.apply(x, weight, bias, relu, leaky_slope, skip,
                          int(bool(grad_premasked)) | 2 * int(bool(mask_input_grad)), float(input_slope))
