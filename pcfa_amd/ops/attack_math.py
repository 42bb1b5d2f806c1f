"""Attack math: box transform, extract_deltas(_joint), loss_delta_constraint and the metric helpers
(helper_functions/own_models.py:62-85, attack_PCFA.py:20-37, helper_functions/losses.py), RAFT's input normalisation."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream, current_lane


# --------------------------------------------------------------------------- #
# attack math
# --------------------------------------------------------------------------- #
class _Pm1Pair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image1, image2):
        _dev(image1, image2)
        if image1.shape != image2.shape:
            raise ValueError("pm1_pair: shapes differ: %s vs %s" % (tuple(image1.shape), tuple(image2.shape)))
        a, b = image1.contiguous(), image2.contiguous()
        B = a.shape[0]
        n = a.numel() // B
        pair = torch.empty((2 * B,) + tuple(a.shape[1:]), device=a.device, dtype=torch.float32)
        cx = torch.empty_like(a)
        _call("pcfa_pm1_pair_fwd", _ptr(a), _ptr(b), _ptr(pair), _ptr(cx), B, n)
        ctx.set_materialize_grads(False)    # an unused output hands None to the backward, not a zero tensor
        ctx.dims = (B, n, tuple(a.shape))
        return pair, cx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gpair, gctx):
        B, n, shape = ctx.dims
        if gpair is None and gctx is None:
            return None, None
        if gpair is None:   # only the context-encoder branch carries gradient
            gpair = torch.zeros((2 * B,) + shape[1:], device=gctx.device, dtype=torch.float32)
        gpair = gpair.contiguous()
        gctx = None if gctx is None else gctx.contiguous()
        ga = torch.empty(shape, device=gpair.device, dtype=torch.float32)
        gb = torch.empty_like(ga)
        _call("pcfa_pm1_pair_bwd", _ptr(gpair), _ptr(gctx), _ptr(ga), _ptr(gb), B, n)
        return ga, gb


def pm1_pair(image1, image2):
    """(cat([n(image1), n(image2)]), n(image1)) with n(x) = 2 * (x / 255.0) - 1.0 (raft.py:88-89): the feature encoder's
    batch and the context encoder's input in one launch per direction."""
    return _Pm1Pair.apply(image1, image2)


class _BoxTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, delta, cov, eps_box, scale):
        _dev(image, delta)
        lib = _hip.load()
        img = image.contiguous()
        d = None if delta is None else delta.contiguous()
        B = img.shape[0]
        n = img.numel() // B
        if d is not None and d.numel() != n:
            raise ValueError("delta must broadcast over the batch: %s vs %s" % (tuple(d.shape), tuple(img.shape)))
        out = torch.empty_like(img)
        _call("pcfa_box_transform_fwd", _ptr(img), _ptr(d), _ptr(out), B, n, int(cov), float(eps_box),
                                              float(scale))
        ctx.args = (B, n, int(cov), float(eps_box), float(scale))
        ctx.delta_shape = None if delta is None else delta.shape
        ctx.save_for_backward(img, d)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        img, d = ctx.saved_tensors
        lib = _hip.load()
        B, n, cov, eps, scale = ctx.args
        g = grad_out.contiguous()
        need_img, need_delta = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and d is not None
        gi = torch.empty_like(img) if need_img else None
        gd = torch.empty(ctx.delta_shape, device=img.device, dtype=torch.float32) if need_delta else None
        _call("pcfa_box_transform_bwd", _ptr(img), _ptr(d), _ptr(g), _ptr(gi), _ptr(gd), B, n, cov, eps,
                                              scale)
        return gi, gd, None, None, None


def box_transform(image, delta=None, change_of_variables=False, eps_box=0., scale=1.):
    """clamp(cov(image + delta), 0, 1) * scale -- ScaledInputModel.forward prologue for one image."""
    return _BoxTransform.apply(image, delta, change_of_variables, eps_box, scale)


class _ExtractDeltas(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nw_input, image, cov, eps_box):
        _dev(nw_input, image)
        lib = _hip.load()
        w = nw_input.contiguous()
        img = image.contiguous()
        out = torch.empty_like(w)
        _call("pcfa_extract_deltas_fwd", _ptr(w), _ptr(img), _ptr(out), w.numel(), int(cov),
                                               float(eps_box))
        ctx.args = (int(cov), float(eps_box))
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, grad_delta):
        (w,) = ctx.saved_tensors
        lib = _hip.load()
        g = grad_delta.contiguous()
        gw = torch.empty_like(w)
        _call("pcfa_extract_deltas_bwd", _ptr(w), _ptr(g), _ptr(gw), w.numel(), ctx.args[0], ctx.args[1])
        return gw, None, None, None


class _ExtractDeltasJoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nw_delta, images_max, images_min):
        _dev(nw_delta, images_max, images_min)
        lib = _hip.load()
        nd, mx, mn = nw_delta.contiguous(), images_max.contiguous(), images_min.contiguous()
        out = torch.empty_like(nd)
        _call("pcfa_extract_deltas_joint_fwd", _ptr(nd), _ptr(mx), _ptr(mn), _ptr(out), nd.numel())
        ctx.save_for_backward(nd, mx, mn)
        return out

    @staticmethod
    def backward(ctx, grad_delta):
        nd, mx, mn = ctx.saved_tensors
        lib = _hip.load()
        g = grad_delta.contiguous()
        gnd = torch.empty_like(nd)
        _call("pcfa_extract_deltas_joint_bwd", _ptr(nd), _ptr(mx), _ptr(mn), _ptr(g), _ptr(gnd), nd.numel())
        return gnd, None, None


def extract_deltas(nw_input1, nw_input2, image1, image2, boxconstraint, eps_box=0.):
    cov = boxconstraint in ['change_of_variables']
    return (_ExtractDeltas.apply(nw_input1, image1, cov, eps_box),
            _ExtractDeltas.apply(nw_input2, image2, cov, eps_box))


def extract_deltas_joint(nw_delta, images_max, images_min):
    delta = _ExtractDeltasJoint.apply(nw_delta, images_max, images_min)
    return delta, delta


_WS = {}


def _workspace(device):
    """Reduction scratch of the loss / metric kernels (32 KB), one per (device, stream, lane), allocated once.
    While a hipGraph is being captured the capture stream reuses a buffer that was allocated OUTSIDE any capture
    (every capture in this package is preceded by eager warm-up calls on the same device), so no scratch comes from --
    and pins -- a graph's private memory pool; the kernels of one closure are stream-ordered on one stream at a time."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    ln = current_lane()
    key = (idx, None if capturing else torch.cuda.current_stream().cuda_stream, ln)
    ws = _WS.get(key)
    if ws is None and capturing:
        ws = next((w for (d, s_, l_), w in _WS.items() if d == idx and s_ is not None and l_ == ln), None)
    if ws is None:
        nbytes = _hip.load().pcfa_flow_loss_workspace_bytes()
        ws = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
        _WS[key] = ws
    return ws


def _flow4(t):
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dim() != 4 or t.shape[1] != 2:
        raise ValueError("The flow tensors do not have a valid number of dimensions "
                         "(either [b,2,M,N] or [2,M,N]). Here: %s" % str(t.size()))
    return t


class _LossDeltaConstraint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, delta1, delta2, delta_bound, mu, f_type, batch_sums=None):
        _dev(pred, target, delta1, delta2)
        lib = _hip.load()
        p, t = _flow4(pred), _flow4(target)
        if p.shape != t.shape:
            raise ValueError("pred/target shape mismatch: %s vs %s" % (tuple(p.shape), tuple(t.shape)))
        d1, d2 = delta1.contiguous(), delta2.contiguous()
        B, _, H, W = p.shape
        scal = torch.empty(8, device=p.device, dtype=torch.float32)
        ft = _hip.PCFA_LOSS[f_type]
        _call("pcfa_flow_loss_fwd", _ptr(p), _hip.strides4(p), _ptr(t), _hip.strides4(t), B, H, W,
                                          _ptr(d1), d1.numel(), _ptr(d2), d2.numel(), float(delta_bound),
                                          float(mu), ft, _ptr(scal), _ptr(_workspace(p.device)))
        ctx.sim_scale = 1
        if batch_sums is not None and f_type == "cosim":
            # this rank holds a slice of the batch: the three sums of f_cosim (losses.py:88) become the sums over the
            # global batch before anything reads them (12-byte all-reduce), the scalars are re-derived from them in
            # the kernel's own operation order, and the backward kernel reads the global sums from `scal`
            ctx.sim_scale = int(batch_sums(scal[3:6]))
            sim = 1.0 - scal[3] / torch.sqrt(scal[4]) * torch.sqrt(scal[5])
            scal[1] = sim
            scal[0] = sim + float(mu) * torch.clamp_min(scal[6], 0.0)
        ctx.joint = d1.data_ptr() == d2.data_ptr() and d1.numel() == d2.numel()
        ctx.args = (B, H, W, float(mu), ft)
        ctx.pred_shape = pred.shape
        ctx.save_for_backward(p, t, d1, d2, scal)
        return scal[0].clone()

    @staticmethod
    def backward(ctx, grad_loss):
        p, t, d1, d2, scal = ctx.saved_tensors
        lib = _hip.load()
        B, H, W, mu, ft = ctx.args
        gl = grad_loss.contiguous().reshape(1)
        need_p, need_d1, need_d2 = ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        gp = torch.empty((B, 2, H, W), device=p.device, dtype=torch.float32) if need_p else None
        gd1 = torch.empty_like(d1) if need_d1 else None
        gd2 = torch.empty_like(d2) if (need_d2 and not ctx.joint) else None
        if ctx.joint and need_d2 and gd1 is None:
            gd1 = torch.empty_like(d1)
        _call("pcfa_flow_loss_bwd", _ptr(p), _hip.strides4(p), _ptr(t), _hip.strides4(t), B, H, W,
                                          _ptr(d1), d1.numel(), _ptr(d2), d2.numel(), mu, ft, 0,
                                          _ptr(scal), _ptr(gl), _ptr(gp), _ptr(gd1), _ptr(gd2))
        if gp is not None:
            gp = gp.reshape(ctx.pred_shape)
            if ctx.sim_scale != 1:   # gradients are AVERAGED over ranks afterwards; the similarity term is a sum
                gp.mul_(float(ctx.sim_scale))
        if ctx.joint:
            # extract_deltas_joint hands the SAME tensor in twice (attack_PCFA.py:37): autograd adds the
            # two slots, which reproduces the reference's d/d(delta) of |delta|^2 + |delta|^2.
            return gp, None, (gd1 if need_d1 else None), (gd1 if need_d2 else None), None, None, None, None
        return gp, None, gd1, gd2, None, None, None, None


def loss_delta_constraint(pred, target, delta1, delta2, device=None, delta_bound=0.001, mu=100., f_type="aee",
                          batch_sums=None):
    """helper_functions/losses.py:200-230 (device argument kept for signature compatibility).
    batch_sums: multi-rank universal attack with cosim only -- all-reduces [p.t, p.p, t.t] in place, returns the
    number of ranks (see UniversalAttack); None everywhere else."""
    if f_type not in _hip.PCFA_LOSS:
        raise NotImplementedError(
            "The requested loss type %s does not exist. Please choose one of 'aee', 'mse' or 'cosim'" % f_type)
    return _LossDeltaConstraint.apply(pred, target, delta1, delta2, delta_bound, mu, f_type, batch_sums)


def get_loss(f_type, pred, target):
    """helper_functions/losses.py:145-174: the similarity term alone (penalty weight 0 on a dummy perturbation)."""
    z = torch.zeros(4, device=pred.device, dtype=torch.float32)
    return _LossDeltaConstraint.apply(pred, target, z, z, 1.0, 0.0, f_type)


def relu_penalty(delta1, delta2, device=None, delta_bound=0.001):
    """helper_functions/losses.py:177-197: relu(mean(delta^2) - delta_bound^2), differentiable.  Runs the fused loss
    kernels with mu = 1 on a zero flow pair, whose MSE similarity term is exactly 0 (value and gradient)."""
    z = torch.zeros((1, 2, 1, 1), device=delta1.device, dtype=torch.float32)
    return _LossDeltaConstraint.apply(z, z, delta1, delta2, delta_bound, 1.0, "mse")


def two_norm_avg_delta_squared(delta1, delta2):
    """helper_functions/losses.py:110-126: (sum d1^2 + sum d2^2) / (n1 + n2), differentiable (= the penalty with a
    zero bound: the mean square is never negative, so the relu is the identity)."""
    return relu_penalty(delta1, delta2, None, 0.0)


def avg_epe(flow1, flow2):
    """helper_functions/losses.py:3-30 (metric use: no gradient)."""
    _dev(flow1, flow2)
    lib = _hip.load()
    a, b = _flow4(flow1.detach()), _flow4(flow2.detach())
    if a.shape != b.shape:
        raise ValueError("flow shape mismatch")
    B, _, H, W = a.shape
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    _call("pcfa_avg_epe", _ptr(a), _hip.strides4(a), _ptr(b), _hip.strides4(b), B, H, W, _ptr(out),
                                _ptr(_workspace(a.device)))
    return out[0]


def sum_squares(x):
    _dev(x)
    lib = _hip.load()
    xc = x.detach().contiguous()
    out = torch.empty(1, device=xc.device, dtype=torch.float32)
    _call("pcfa_sum_squares", _ptr(xc), xc.numel(), _ptr(out), _ptr(_workspace(xc.device)))
    return out[0]


def two_norm_avg(x):
    """helper_functions/losses.py:129-142."""
    return torch.sqrt(sum_squares(x)) / (torch.numel(x) ** 0.5)


def two_norm_avg_delta(delta1, delta2):
    """helper_functions/losses.py:91-107."""
    sqrt_numels = (torch.numel(delta1) + torch.numel(delta2)) ** 0.5
    return torch.sqrt(sum_squares(delta1) + sum_squares(delta2)) / sqrt_numels
