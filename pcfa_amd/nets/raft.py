"""RAFT (Teed & Deng 2020) for the PCFA hot path, written against pcfa_amd.ops.

Parameter names are those of the public RAFT checkpoints (raft-sintel.pth), so
`load_state_dict` works on them after stripping DataParallel's "module."
prefix.  Behavioural reference (cv-stuttgart/PCFA):
    models/raft/raft.py:24-144      network wiring, 12 refinement iterations
    models/raft/extractor.py:6-58,118-192   residual encoder
    models/raft/update.py:6-16,33-60,79-136 motion encoder, SepConvGRU, heads
    models/raft/corr.py:12-60       -> pcfa_amd.ops.get().CorrBlock (HIP)

Differences that do not change results:
  * convolutions / norms run on MIOpen through torch; the all-pairs volume,
    its pyramid and the lookups run in the hand-written HIP kernels;
  * in test_mode the convex-upsampling mask head and upsample_flow are only
    evaluated for the last iteration (the reference computes all 12 and drops
    11, raft.py:131-142);
  * only the full-size model ("small": false in models/_config/raft_config.json).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import cfg


def _norm(kind, planes):
    if kind == 'batch':
        return nn.BatchNorm2d(planes)
    if kind == 'instance':
        return nn.InstanceNorm2d(planes)
    if kind == 'group':
        return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
    if kind == 'none':
        return nn.Sequential()
    raise ValueError("unknown norm_fn %r" % kind)


def _fold_batchnorm(conv, bn, cache, tag):
    """(W', b') with  bn(conv(x)) == conv(x; W') + b'  for an eval-mode BatchNorm2d: W' = W*s, b' = (b - mean)*s + beta,
    s = gamma / sqrt(var + eps).  Cached per (weights, statistics) version."""
    tensors = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (conv.weight.device,)
    hit = cache.get(tag)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            s = torch.rsqrt(bn.running_var + bn.eps)
            if bn.weight is not None:
                s = s * bn.weight
            b = -bn.running_mean if conv.bias is None else conv.bias - bn.running_mean
            b = b * s
            if bn.bias is not None:
                b = b + bn.bias
            hit = (key, (conv.weight * s[:, None, None, None]).contiguous(), b.contiguous())
        cache[tag] = hit
    return hit[1], hit[2]


# Build switches (fused lookup, stride-2 kernels, deferred ReLU masks, two-stream encoders ...) come from the frozen
# pcfa_amd.config.Config the model was built with: cfg(module).<switch>.  The two-stream encoders (overlap_encoders) stay
# opt-in: measured 13.10 -> 12.95 ms per captured closure -- and one hard failure: with several closures captured in one
# process a replay of a two-branch graph crashed inside hipGraphLaunch.


def _conv_norm(conv, norm, x, relu, cache, tag, skip=False, grad_premasked=False, mask_input_grad=False):
    """relu?(norm(conv(x))) of the residual encoders (extractor.py:23-58,161-177).

    Frozen-weight fast paths (what the attack runs), identical to the reference in exact arithmetic:
      * eval-mode BatchNorm (context encoder) is an affine map per channel: folded into the convolution's weights,
        the remaining bias (+ ReLU) applied by one fused pass -- no normalisation kernels in forward or backward;
      * InstanceNorm without affine parameters (feature encoder) subtracts the per-plane mean, which cancels the
        convolution's bias exactly: the bias add (a full pass over the activation) is skipped."""
    frozen = not (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad))
    if (frozen and isinstance(norm, nn.BatchNorm2d) and not norm.training and norm.track_running_stats
            and not (norm.affine and (norm.weight.requires_grad or norm.bias.requires_grad))):
        w, b = _fold_batchnorm(conv, norm, cache, tag)
        if _is_plain3x3(conv):
            return ops.get().conv3x3(x, w, b, relu, skip=skip, grad_premasked=grad_premasked,
                                     mask_input_grad=mask_input_grad)
        assert not (skip or grad_premasked or mask_input_grad)
        if _is_stride2(conv, x, cfg(conv)):
            return ops.get().conv_s2(x, w, b, relu, own_bwd=cfg(conv).conv_s2_bwd)
        if relu:
            return ops.get().bias_relu(conv._conv_forward(x, w, None), b)
        return conv._conv_forward(x, w, b)
    if (frozen and isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats):
        xs = None
        if _is_plain3x3(conv):
            y = ops.get().conv3x3(x, conv.weight, None, False, skip=skip)
            if skip:
                y, xs = y
        elif _is_stride2(conv, x, cfg(conv)):
            assert not skip
            y = ops.get().conv_s2(x, conv.weight, None, False, own_bwd=cfg(conv).conv_s2_bwd)
        else:
            assert not skip
            y = conv._conv_forward(x, conv.weight, None)
        # normalisation + ReLU in two streaming launches per direction instead of the library's 3 + 2 passes
        y = ops.get().instance_norm_relu(y, norm.eps, relu)
        return (y, xs) if skip else y
    assert not skip
    y = norm(conv(x))
    return F.relu(y, inplace=True) if relu else y


def _is_stride2(conv, x, conf):
    """A frozen k x k / stride-2 / padding k//2 convolution ops.conv_s2 covers (the stem, the residual blocks' entry)."""
    k = conv.kernel_size[0]
    return (conf.conv_s2 and conv.kernel_size == (k, k) and conv.stride == (2, 2) and conv.padding == (k // 2, k // 2)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and ops.get().conv_s2_supported(x, conv.weight))


def _can_skip(conv, norm):
    """_conv_norm(.., skip=True) is available: a frozen plain 3x3 convolution on one of the two fast paths."""
    frozen = not (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad))
    if not (frozen and _is_plain3x3(conv)):
        return False
    if isinstance(norm, nn.BatchNorm2d):
        return (not norm.training and norm.track_running_stats
                and not (norm.affine and (norm.weight.requires_grad or norm.bias.requires_grad)))
    return isinstance(norm, nn.InstanceNorm2d) and not norm.affine and not norm.track_running_stats


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            # registered twice (norm3 and downsample.1) like the checkpoints expect
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)
        self._fold_cache = {}

    def _entry_pair(self, x):
        """(relu(norm1(conv1(x))), norm3(downsample[0](x))) of a stride-2 block through ONE fused launch per direction
        (ops.conv_s2_ds), or None when the fast path does not apply."""
        conv1, convd, n1, nd = self.conv1, self.downsample[0], self.norm1, self.downsample[1]
        o = ops.get()
        if not (cfg(self).fused_downsample and _all_frozen(self) and conv1.stride == (2, 2) and convd.stride == (2, 2)
                and convd.kernel_size == (1, 1) and convd.padding == (0, 0) and _is_stride2(conv1, x, cfg(self))
                and o.conv_s2_ds_supported(x, conv1.weight, convd.weight)):
            return None
        if all(isinstance(n, nn.BatchNorm2d) and not n.training and n.track_running_stats for n in (n1, nd)):
            w1, b1 = _fold_batchnorm(conv1, n1, self._fold_cache, "1")
            wd, bd = _fold_batchnorm(convd, nd, self._fold_cache, "d")
            if not o.conv_s2_ds_supported(x, w1, wd):
                return None
            return o.conv_s2_ds(x, w1, wd, b1, bd, relu=True)
        if all(isinstance(n, nn.InstanceNorm2d) and not n.affine and not n.track_running_stats for n in (n1, nd)):
            y, xd = o.conv_s2_ds(x, conv1.weight, convd.weight)      # the biases cancel in the instance norms
            return o.instance_norm_relu(y, n1.eps, True), o.instance_norm_relu(xd, nd.eps, False)
        return None

    def forward(self, x):
        pair = self._entry_pair(x) if self.downsample is not None else None
        if pair is not None:
            y, xd = pair
            y = _conv_norm(self.conv2, self.norm2, y, True, self._fold_cache, "2")
            return ops.get().add_relu(xd, y)
        if self.downsample is None and _all_frozen(self) and _can_skip(self.conv1, self.norm1):
            # the block input feeds conv1 and the residual sum: conv1's data-gradient kernel adds the residual path's
            # gradient in its epilogue (one autograd `add` over the activation less per block)
            y, x = _conv_norm(self.conv1, self.norm1, x, True, self._fold_cache, "1", skip=True,
                              grad_premasked=cfg(self).defer_relu and isinstance(self.norm1, nn.BatchNorm2d) and isinstance(self.norm2, nn.BatchNorm2d)
                              and _can_skip(self.conv2, self.norm2))
        else:
            y = _conv_norm(self.conv1, self.norm1, x, True, self._fold_cache, "1")
        # folded-BatchNorm blocks (context encoder) are conv + ReLU chains: conv2's data-gradient kernel applies conv1's
        # ReLU mask, and the block's output ReLU backward masks conv2's gradient in the same pass (2 launches less)
        chain = (cfg(self).defer_relu and _all_frozen(self) and isinstance(self.norm2, nn.BatchNorm2d) and _can_skip(self.conv2, self.norm2)
                 and _can_skip(self.conv1, self.norm1) and self.downsample is None)
        y = _conv_norm(self.conv2, self.norm2, y, True, self._fold_cache, "2", grad_premasked=chain, mask_input_grad=chain)
        if self.downsample is not None:
            x = _conv_norm(self.downsample[0], self.downsample[1], x, False, self._fold_cache, "d")
        if _all_frozen(self):
            # one pass instead of add + ReLU; one backward kernel for both operands
            return ops.get().add_relu(x, y, b_is_relu=chain)
        return self.relu(x + y)


class BasicEncoder(nn.Module):
    """1/8-resolution feature / context encoder."""

    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = nn.GroupNorm(num_groups=8, num_channels=64) if norm_fn == 'group' else _norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = self._stage(64, 64, 1)
        self.layer2 = self._stage(64, 96, 2)
        self.layer3 = self._stage(96, 128, 2)
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        self._fold_cache = {}
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _stage(self, cin, cout, stride):
        return nn.Sequential(ResidualBlock(cin, cout, self.norm_fn, stride=stride),
                             ResidualBlock(cout, cout, self.norm_fn, stride=1))

    def forward(self, x, split=None):
        """x: a tensor, a pair of equally shaped tensors (run as one batch, results split again), or with `split` = n a
        batch that already holds both images ([:n], [n:])."""
        pair = isinstance(x, (tuple, list)) or split is not None
        if split is not None:
            n = split
        elif pair:
            n = x[0].shape[0]
            x = torch.cat(x, dim=0)
        x = _conv_norm(self.conv1, self.norm1, x, True, self._fold_cache, "1")
        x = self.layer3(self.layer2(self.layer1(x)))
        x = _conv1x1(self.conv2, x, cfg(self))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        if pair:
            x = torch.split(x, [n, n], dim=0)
        return x


def _all_frozen(module):
    return not any(p.requires_grad for p in module.parameters())


def _conv_nobias(conv, x):
    """conv(x) without its bias: the bias add is folded into the fused HIP kernel that consumes the result."""
    return conv._conv_forward(x, conv.weight, None)


def _is_plain3x3(conv, min_out=16):
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.out_channels >= min_out)


def _is_plain1x1(conv):
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1)


def _conv1x1(conv, x, config):
    """conv(x) for a 1x1 layer: the library's convolution, or -- Config.conv1x1 = "hip", frozen weights, GPU -- the package's
    own product (ops.conv1x1)."""
    if config.conv1x1 == "hip" and x.is_cuda and _is_plain1x1(conv) and _frozen_conv(conv):
        return ops.get().conv1x1(x, conv.weight, conv.bias)
    return conv(x)


def _frozen_conv(conv):
    return not (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad))


def _conv_relu(conv, x):
    """relu(conv(x)).  Frozen 3x3 / stride 1 convolutions run as Winograd F(2x2,3x3) on the fp32 matrix cores with bias
    and ReLU in the epilogue (ops.conv3x3); everything else: library convolution, bias + ReLU in one fused pass."""
    if _is_plain3x3(conv) and _frozen_conv(conv):
        return ops.get().conv3x3(x, conv.weight, conv.bias, True)
    o = ops.get()
    k = conv.kernel_size[0]
    if (_frozen_conv(conv) and not x.requires_grad and conv.kernel_size == (k, k) and conv.stride == (1, 1)
            and conv.padding == (k // 2, k // 2) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and (conv.in_channels, k) in o.FEWIN_SHAPES):
        # convf1 on the detached flow: 2 input channels, no data gradient -- one streaming launch, bias + ReLU fused
        return o.conv_fewin(x, conv.weight, conv.bias, True)
    return o.bias_relu(_conv_nobias(conv, x), conv.bias)


def _predict_flow(conv, x):
    """conv(x) for a flow-prediction layer (3x3, <= 4 output channels): frozen weights stream through
    ops.conv3x3_fewout instead of a library convolution padded to a matrix-core tile."""
    if (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.out_channels <= 4 and _frozen_conv(conv)):
        return ops.get().conv3x3_fewout(x, conv.weight, conv.bias)
    return conv(x)


class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return _predict_flow(self.conv2, _conv_relu(self.conv1, x))


class SepConvGRU(nn.Module):
    """Two GRU half-steps with 1x5 then 5x1 gates (update.py:33-60):
        z = sigmoid(convz(hx)); r = sigmoid(convr(hx)); q = tanh(convq([r*h, x])); h' = (1-z)*h + z*q,   hx = [h, x].

    Frozen-weight fast path (what the attack runs): the gate convolutions are linear in their input channels and
    x = [inp | rest] starts with the context features `inp`, which do not change over the refinement
    iterations.  `precompute(inp)` evaluates conv(inp, W[:, inp-slice]) + bias once per forward; every iteration
    then only convolves [h | rest] (2/3 of the channels for RAFT) and the fused HIP gate kernels add the cached
    part.  z and r share their input, so they run as one convolution with stacked weights.  Identical to the
    reference in exact arithmetic; in fp32 it regroups the channel sum (covered by the closure parity tests).
    """

    def __init__(self, hidden_dim=128, input_dim=192 + 128, const_dim=128):
        super().__init__()
        c = hidden_dim + input_dim
        self.hidden_dim, self.const_dim = hidden_dim, const_dim
        self.convz1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(c, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(c, hidden_dim, (5, 1), padding=(2, 0))
        self._split_cache = {}

    # ---- reference formulation (also used when the weights are being trained) ----------------
    @staticmethod
    def _half(h, x, convz, convr, convq):
        o = ops.get()
        hx = torch.cat([h, x], dim=1)
        z, rh = o.gru_gates(_conv_nobias(convz, hx), _conv_nobias(convr, hx), h, convz.bias, convr.bias)
        return o.gru_update(z, _conv_nobias(convq, torch.cat([rh, x], dim=1)), h, convq.bias)

    def forward(self, h, x):
        h = self._half(h, x, self.convz1, self.convr1, self.convq1)
        return self._half(h, x, self.convz2, self.convr2, self.convq2)

    # ---- frozen-weight fast path -------------------------------------------------------------
    def frozen(self):
        return not any(p.requires_grad for p in self.parameters())

    def _split(self, tag, convs):
        """(W_dyn over [h | rest], W_const over inp, bias) for the output-stacked convolutions `convs`."""
        key = tuple((c.weight.data_ptr(), c.weight._version) for c in convs) + (convs[0].weight.device,)
        hit = self._split_cache.get(tag)
        if hit is None or hit[0] != key:
            hd, cd = self.hidden_dim, self.const_dim
            with torch.no_grad():
                w = torch.cat([c.weight for c in convs], dim=0)
                b = torch.cat([c.bias for c in convs], dim=0).contiguous()
                w_dyn = torch.cat([w[:, :hd], w[:, hd + cd:]], dim=1).contiguous()
                w_const = w[:, hd:hd + cd].contiguous()
            hit = (key, w_dyn, w_const, b)
            self._split_cache[tag] = hit
        return hit[1:]

    def precompute(self, inp):
        """Per forward: the contribution of the constant context features to the six gate pre-activations."""
        ctx = {}
        for tag, convs in (("zr1", (self.convz1, self.convr1)), ("q1", (self.convq1,)),
                           ("zr2", (self.convz2, self.convr2)), ("q2", (self.convq2,))):
            w_dyn, w_const, b = self._split(tag, convs)
            # the same (1,5)/(5,1) kernel as the per-iteration gate convolutions, here on the context features alone:
            # 230 us against 340 us in the library for the four of them (forward + data gradient, r03 probe)
            ctx[tag] = (w_dyn, ops.get().sepconv5(inp, None, w_const) + b.view(1, -1, 1, 1))
        return ctx

    def per_iteration(self, ctx, iters):
        """precompute()'s context for each of `iters` refinement iterations: the cached pre-activations are read by
        every iteration, so their gradient is a sum of `iters` terms -- ops.fanout hands out one alias per iteration
        and adds the gradients with one launch per tensor (autograd would accumulate them pairwise)."""
        fans = {tag: ops.get().fanout(p, iters) for tag, (_, p) in ctx.items()}
        return [{tag: (ctx[tag][0], fans[tag][i]) for tag in ctx} for i in range(iters)]

    def step(self, h, ctx, rest, rest_relu_channels=0):
        """One GRU update given precompute()'s context; `rest` = the per-iteration part of x (motion features).
        rest_relu_channels: see ops.gru_step (the leading channels of `rest` whose ReLU backward this node applies)."""
        # one autograd node per update: sepconv5 reads [h | rest] in place (no torch.cat, no im2col) and the
        # backward accumulates the gradients of h and rest inside the kernels that produce them
        return ops.get().gru_step(h, rest, tuple(ctx[zr] + ctx[q] for zr, q in (("zr1", "q1"), ("zr2", "q2"))),
                                  rest_relu_channels)


class LookupRef:
    """A correlation lookup that has not been evaluated yet: the motion encoder asks for relu(convc1(lookup)) and the
    operator table may serve both in one launch (CorrBlock.lookup_conv_relu: the [324][Q] tensor never exists);
    anything else materialises the lookup and runs the layers one by one, as the reference does
    (models/raft/raft.py:123-124, update.py:79-93)."""

    def __init__(self, corr_fn, coords):
        self.corr_fn, self.coords = corr_fn, coords

    def tensor(self):
        return self.corr_fn(self.coords)

    def conv_relu(self, conv):
        fused = getattr(self.corr_fn, "lookup_conv_relu", None)
        if fused is not None and _frozen_conv(conv) and cfg(conv).fused_lookup:
            out = fused(self.coords, conv.weight, conv.bias, True)
            if out is not None:
                return out
        return _conv_relu(conv, self.tensor())


class BasicMotionEncoder(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    RELU_CHANNELS = 128 - 2   # leading channels of the result that are ReLU outputs (the rest is the flow)

    def can_defer_relu(self, flow):
        """True when forward(.., defer_relu=True) is available: the in-place concatenation path below."""
        return (flow.shape[0] == 1 and _all_frozen(self)
                and all(_is_plain3x3(c) for c in (self.convc2, self.convf2, self.conv)))

    def forward(self, flow, corr, defer_relu=False):
        """`corr`: the lookup tensor [B,324,H,W] or a LookupRef (lookup + convc1 may then run as one kernel).
        defer_relu: the caller's ONLY consumer of the result is ops.gru_step(.., rest_relu_channels=RELU_CHANNELS), which
        applies the last layer's ReLU backward itself; the inner ReLUs are then differentiated by the data-gradient
        kernels of the layers they feed (ops.conv3x3_cat flags) -- two elementwise launches less per iteration."""
        cor1 = corr.conv_relu(self.convc1) if isinstance(corr, LookupRef) else _conv_relu(self.convc1, corr)
        if self.can_defer_relu(flow):
            # the 3x3 convolutions write their channel blocks of the concatenated tensors in place (no torch.cat pass)
            o = ops.get()
            flo1 = _conv_relu(self.convf1, flow)
            cf = o.conv3x3_cat([(cor1, self.convc2.weight, self.convc2.bias), (flo1, self.convf2.weight, self.convf2.bias)],
                               grad_premasked=defer_relu)
            return o.conv3x3_cat([(cf, self.conv.weight, self.conv.bias)], (flow,), grad_premasked=defer_relu,
                                 mask_input_grads=defer_relu)
        if defer_relu:
            raise RuntimeError("BasicMotionEncoder: defer_relu needs the in-place path (check can_defer_relu first)")
        cor = _conv_relu(self.convc2, cor1)
        flo = _conv_relu(self.convf2, _conv_relu(self.convf1, flow))
        out = _conv_relu(self.conv, torch.cat([cor, flo], dim=1))
        return torch.cat([out, flow], dim=1)


def _mask_head():
    return nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1, padding=0))


def mask_logits(head, net, cache):
    """.25 * head(net) (update.py:136-139: "scale mask to balance gradients").  Frozen weights: the 3x3 convolution + ReLU
    run on ops.conv3x3 like every other 3x3 layer, and the factor is folded into the 1x1 convolution's weight and bias --
    a power of two commutes with every fp32 rounding, so the result is bit-identical and the two passes over the
    [N,576,H,W] logits (forward and backward) disappear."""
    c0, c2 = head[0], head[2]
    if not (_frozen_conv(c0) and _frozen_conv(c2) and _is_plain3x3(c0)):
        return .25 * head(net)
    key = (c2.weight.data_ptr(), c2.weight._version, c2.bias.data_ptr(), c2.bias._version)
    if cache.get("key") != key:
        with torch.no_grad():
            cache["key"], cache["w"], cache["b"] = key, (.25 * c2.weight).contiguous(), (.25 * c2.bias).contiguous()
    y = _conv_relu(c0, net)
    if cfg(head).conv1x1 == "hip" and y.is_cuda and _is_plain1x1(c2):
        return ops.get().conv1x1(y, cache["w"], cache["b"])
    return c2._conv_forward(y, cache["w"], cache["b"])


class BasicUpdateBlock(nn.Module):
    def __init__(self, corr_levels=4, corr_radius=4, hidden_dim=128):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = _mask_head()
        self._mask_cache = {}

    def forward(self, net, inp, corr, flow, want_mask=True, gru_ctx=None):
        # frozen weights: the GRU node is the motion features' only consumer and differentiates their ReLU itself
        defer = cfg(self).defer_relu and gru_ctx is not None and self.encoder.can_defer_relu(flow)
        motion_features = self.encoder(flow, corr, defer_relu=defer)
        if gru_ctx is not None:   # frozen weights: context-feature part of the gate convolutions hoisted
            net = self.gru.step(net, gru_ctx, motion_features, self.encoder.RELU_CHANNELS if defer else 0)
        else:
            net = self.gru(net, torch.cat([inp, motion_features], dim=1))
        delta_flow = self.flow_head(net)
        mask = mask_logits(self.mask, net, self._mask_cache) if want_mask else None
        return net, mask, delta_flow


def _f32(x):
    """`.float()` of the reference -- except on fp64 tensors, which only the parity arbiter (the CPU port evaluated in
    double precision, tools/parity_arbiter.py) feeds through this file."""
    return x if x.dtype == torch.float64 else x.float()


def coords_grid(batch, ht, wd, device, dtype=torch.float32):
    """[B,2,ht,wd] pixel grid, channel 0 = x, channel 1 = y (models/raft/utils/utils.py:74-77)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).to(dtype)[None].repeat(batch, 1, 1, 1)


def convex_upsample(flow, mask):
    """[N,2,H,W] -> [N,2,8H,8W] by a softmax-weighted 3x3 combination (raft.py:72-83): the operator table's fused
    kernel pair (pcfa_convex_upsample_fwd / _bwd on the GPU; the reference's tensor expression in the oracle)."""
    return ops.get().convex_upsample(flow, mask)


class RAFT(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = dict(args or {})
        if self.args.get("small", False):
            raise NotImplementedError("RAFT-small is not part of the PCFA configurations (raft_config.json: small=false)")
        self.hidden_dim = hdim = 128
        self.context_dim = cdim = 128
        self.args["corr_levels"] = 4
        self.args["corr_radius"] = 4
        self.args.setdefault("dropout", 0)
        self.fnet = BasicEncoder(output_dim=256, norm_fn='instance', dropout=self.args["dropout"])
        self.cnet = BasicEncoder(output_dim=hdim + cdim, norm_fn='batch', dropout=self.args["dropout"])
        self.update_block = BasicUpdateBlock(4, 4, hidden_dim=hdim)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        # both images normalised in one launch: the feature encoder's batch of two and the context encoder's input
        images12, image1 = ops.get().pm1_pair(image1, image2)
        hdim, cdim = self.hidden_dim, self.context_dim

        side = None
        if cfg(self).overlap_encoders and image1.is_cuda and hasattr(ops.get(), "side_stream"):
            # the context encoder (one image) runs beside the feature encoder (two images): independent until the GRU,
            # and their small-map layers each leave most of the chip idle.  A second stream forks here and joins below;
            # under capture it becomes a parallel branch of the hipGraph, autograd runs each branch's backward on its
            # forward stream.
            main = torch.cuda.current_stream()
            side = ops.get().side_stream(image1.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cnet_out = self.cnet(image1)
        fmap1, fmap2 = self.fnet(images12, split=image1.shape[0])
        corr_fn = ops.get().CorrBlock(_f32(fmap1), _f32(fmap2), num_levels=self.args["corr_levels"],
                                      radius=self.args["corr_radius"], bwd_windows=cfg(self).pyramid_bwd_windows)
        if side is not None:
            main.wait_stream(side)
        else:
            cnet_out = self.cnet(image1)

        net, inp = torch.split(cnet_out, [hdim, cdim], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)

        N, _, H, W = image1.shape
        coords0 = coords_grid(N, H // 8, W // 8, image1.device, _f32(image1).dtype)
        coords1 = coords_grid(N, H // 8, W // 8, image1.device, _f32(image1).dtype)
        if flow_init is not None:
            coords1 = coords1 + flow_init

        gru = self.update_block.gru
        gru_ctx = gru.per_iteration(gru.precompute(inp), iters) if gru.frozen() else None
        flow_predictions = []
        flow_up = None
        flow_cur = coords1 - coords0
        for itr in range(iters):
            coords1 = coords1.detach()  # the lookup gets no coordinate gradient (raft.py:122-123)
            corr = LookupRef(corr_fn, coords1)   # evaluated inside the motion encoder (fused with convc1 where possible)
            flow = flow_cur.detach()             # = coords1 - coords0 of the detached coordinates
            need_up = (not test_mode) or itr == iters - 1
            net, up_mask, delta_flow = self.update_block(net, inp, corr, flow, want_mask=need_up,
                                                         gru_ctx=None if gru_ctx is None else gru_ctx[itr])
            # coords1 + delta_flow and the new flow in one launch (the reference: an add here, a subtract per use)
            coords1, flow_cur = ops.get().flow_step(coords1, delta_flow, coords0)
            if need_up:
                flow_up = convex_upsample(flow_cur, up_mask)
                flow_predictions.append(flow_up)
        if test_mode:
            return flow_cur, flow_up
        return flow_predictions
