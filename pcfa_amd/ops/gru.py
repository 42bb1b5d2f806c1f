"""SepConvGRU update (models/raft/update.py:33-60) and the small per-iteration operators of RAFT / GMA: gate arithmetic,
the one-node GRU step with fused epilogues, bias + ReLU, fan-out of hoisted tensors, flow step, convex up-sampling
(raft.py:72-83,122-137)."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream
from .conv import _sepconv5_packed, _SepConv5


# --------------------------------------------------------------------------- #
# SepConvGRU gate arithmetic (models/raft/update.py:45-60)
# --------------------------------------------------------------------------- #
def _plane_channels(t):
    return t.shape[-2] * t.shape[-1], t.shape[-3]


class _GruGates(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zc, rc, h, bias_z, bias_r, add_z, add_r):
        _dev(zc, rc, h, bias_z, bias_r, add_z, add_r)
        zc, rc, h = zc.contiguous(), rc.contiguous(), h.contiguous()
        az = None if add_z is None else add_z.contiguous()
        ar = None if add_r is None else add_r.contiguous()
        z, r, rh = torch.empty_like(zc), torch.empty_like(zc), torch.empty_like(zc)
        plane, C = _plane_channels(zc)
        _call("pcfa_gru_gates_fwd", _ptr(zc), _ptr(rc), _ptr(h), _ptr(bias_z), _ptr(bias_r), _ptr(az), _ptr(ar),
              _ptr(z), _ptr(r), _ptr(rh), zc.numel(), plane, C)
        ctx.save_for_backward(z, r, h)
        ctx.has_add = (add_z is not None, add_r is not None)
        return z, rh

    @staticmethod
    def backward(ctx, dz, drh):
        z, r, h = ctx.saved_tensors
        dz = torch.zeros_like(z) if dz is None else dz.contiguous()
        drh = torch.zeros_like(z) if drh is None else drh.contiguous()
        dzc, drc, dh = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _call("pcfa_gru_gates_bwd", _ptr(z), _ptr(r), _ptr(h), _ptr(dz), _ptr(drh), _ptr(dzc), _ptr(drc), _ptr(dh),
              z.numel())
        # the addends enter the pre-activations with weight 1: their gradient IS the pre-activation gradient
        return dzc, drc, dh, None, None, (dzc if ctx.has_add[0] else None), (drc if ctx.has_add[1] else None)


class _GruGatesPacked(torch.autograd.Function):
    """Same arithmetic as _GruGates on ONE convolution output zr = [zc | rc] (channels 0..C-1 and C..2C-1):
    the z and r gate convolutions share their input, so they run as a single convolution with stacked weights;
    the halves are addressed in place (no slicing copies) and the gradient comes back packed as well."""

    @staticmethod
    def forward(ctx, zr, h, bias_zr, add_zr):
        _dev(zr, h, bias_zr, add_zr)
        zr, h = zr.contiguous(), h.contiguous()
        add = None if add_zr is None else add_zr.contiguous()
        B, C2, H, W = zr.shape
        C, plane = C2 // 2, H * W
        z, r, rh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        bz = None if bias_zr is None else bias_zr[:C]
        br = None if bias_zr is None else bias_zr[C:]
        n = C * plane
        for b in range(B):  # per batch item the two halves of zr are contiguous blocks
            o, oz = b * n, b * 2 * n
            _call("pcfa_gru_gates_fwd", _ptr_off(zr, oz), _ptr_off(zr, oz + n), _ptr_off(h, o), _ptr(bz), _ptr(br),
                  _ptr_off(add, oz), _ptr_off(add, oz + n), _ptr_off(z, o), _ptr_off(r, o), _ptr_off(rh, o), n,
                  plane, C)
        ctx.save_for_backward(z, r, h)
        ctx.has_add = add_zr is not None
        return z, rh

    @staticmethod
    def backward(ctx, dz, drh):
        z, r, h = ctx.saved_tensors
        B, C, H, W = z.shape
        n = C * H * W
        dz = torch.zeros_like(z) if dz is None else dz.contiguous()
        drh = torch.zeros_like(z) if drh is None else drh.contiguous()
        dzr = torch.empty((B, 2 * C, H, W), device=z.device, dtype=torch.float32)
        dh = torch.empty_like(z)
        for b in range(B):
            o, oz = b * n, b * 2 * n
            _call("pcfa_gru_gates_bwd", _ptr_off(z, o), _ptr_off(r, o), _ptr_off(h, o), _ptr_off(dz, o),
                  _ptr_off(drh, o), _ptr_off(dzr, oz), _ptr_off(dzr, oz + n), _ptr_off(dh, o), n)
        return dzr, dh, None, (dzr if ctx.has_add else None)


class _GruUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, qc, h, bias_q, add_q):
        _dev(z, qc, h, bias_q, add_q)
        z, qc, h = z.contiguous(), qc.contiguous(), h.contiguous()
        aq = None if add_q is None else add_q.contiguous()
        q, hnew = torch.empty_like(z), torch.empty_like(z)
        plane, C = _plane_channels(z)
        _call("pcfa_gru_update_fwd", _ptr(z), _ptr(qc), _ptr(h), _ptr(bias_q), _ptr(aq), _ptr(q), _ptr(hnew),
              z.numel(), plane, C)
        ctx.save_for_backward(z, q, h)
        ctx.has_add = add_q is not None
        return hnew

    @staticmethod
    def backward(ctx, g):
        z, q, h = ctx.saved_tensors
        g = g.contiguous()
        dz, dqc, dh = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        _call("pcfa_gru_update_bwd", _ptr(z), _ptr(q), _ptr(h), _ptr(g), _ptr(dz), _ptr(dqc), _ptr(dh), z.numel())
        return dz, dqc, dh, None, (dqc if ctx.has_add else None)


class _BiasRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        _dev(x, bias)
        x = x.contiguous()
        out = torch.empty_like(x)
        plane, C = _plane_channels(x)
        _call("pcfa_bias_relu_fwd", _ptr(x), _ptr(bias), _ptr(out), x.numel(), plane, C)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(out)
        _call("pcfa_relu_bwd", _ptr(out), _ptr(g), _ptr(gx), out.numel())
        return gx, None


_GRU_EPILOGUES = os.environ.get("PCFA_GRU_EPILOGUES", "1") != "0"   # A/B switch (tools/dev)


class _GruStep(torch.autograd.Function):
    """One SepConvGRU update (both half-steps, models/raft/update.py:45-60) as ONE autograd node with a hand-ordered
    backward.  Forward = the same kernel sequence as composing sepconv5 / gru_gates_packed / gru_update.  In the
    backward every gradient that autograd would sum with separate elementwise kernels -- h is used three times per
    half-step, the motion features four times per step -- is accumulated in place by the kernel that produces it
    (pcfa_sepconv5_fwd_split with accumulate flags, pcfa_gru_gates_bwd_acc): 7 add launches less per refinement
    iteration.  Arguments: h, rest, then per half-step (w_zr, p_zr, w_q, p_q) with p_* = the pre-activation
    contribution of the constant context features (bias included)."""

    @staticmethod
    def forward(ctx, h, rest, w_zr1, p_zr1, w_q1, p_q1, w_zr2, p_zr2, w_q2, p_q2, rest_relu_channels=0):
        _dev(h, rest, w_zr1, p_zr1, w_q1, p_q1, w_zr2, p_zr2, w_q2, p_q2)
        h, rest = h.contiguous(), rest.contiguous()
        B, C, H, W = h.shape
        Cr = rest.shape[1]
        n, plane = C * H * W, H * W
        new = lambda c: torch.empty((B, c, H, W), device=h.device, dtype=torch.float32)  # noqa: E731
        saved, packs = [], []
        for w_zr, p_zr, w_q, p_q in ((w_zr1, p_zr1, w_q1, p_q1), (w_zr2, p_zr2, w_q2, p_q2)):
            if tuple(w_zr.shape[:2]) != (2 * C, C + Cr) or tuple(w_q.shape[:2]) != (C, C + Cr):
                raise ValueError("gru_step: weights %s / %s do not fit h %s, rest %s"
                                 % (tuple(w_zr.shape), tuple(w_q.shape), tuple(h.shape), tuple(rest.shape)))
            vertical = int(w_zr.shape[2] == 5)
            f_zr, b_zr = _sepconv5_packed(w_zr)
            f_q, b_q = _sepconv5_packed(w_q)
            p_zr, p_q = p_zr.contiguous(), p_q.contiguous()
            z, r, rh, q, hnew = new(C), new(C), new(C), new(C), new(C)
            if C % 32 == 0 and _GRU_EPILOGUES:
                # gate / update arithmetic in the convolutions' epilogues: the pre-activations never reach memory
                _call("pcfa_sepconv5_gru_gates_fwd", _ptr(h), C, _ptr(rest), Cr, _ptr(f_zr), _ptr(p_zr), _ptr(z), _ptr(r),
                      _ptr(rh), B, H, W, vertical)
                _call("pcfa_sepconv5_gru_update_fwd", _ptr(rh), C, _ptr(rest), Cr, _ptr(f_q), _ptr(p_q), _ptr(z), _ptr(h),
                      _ptr(q), _ptr(hnew), B, H, W, vertical)
            else:
                zr, qc = new(2 * C), new(C)
                _call("pcfa_sepconv5_fwd", _ptr(h), C, _ptr(rest), Cr, _ptr(f_zr), _ptr(zr), B, 2 * C, H, W, vertical)
                for b in range(B):  # per batch item the z and r halves of zr are contiguous blocks
                    o, oz = b * n, b * 2 * n
                    _call("pcfa_gru_gates_fwd", _ptr_off(zr, oz), _ptr_off(zr, oz + n), _ptr_off(h, o), None, None,
                          _ptr_off(p_zr, oz), _ptr_off(p_zr, oz + n), _ptr_off(z, o), _ptr_off(r, o), _ptr_off(rh, o),
                          n, plane, C)
                _call("pcfa_sepconv5_fwd", _ptr(rh), C, _ptr(rest), Cr, _ptr(f_q), _ptr(qc), B, C, H, W, vertical)
                _call("pcfa_gru_update_fwd", _ptr(z), _ptr(qc), _ptr(h), None, _ptr(p_q), _ptr(q), _ptr(hnew),
                      z.numel(), plane, C)
            saved += [z, r, q, h]
            packs.append((b_zr, b_q, vertical))
            h = hnew
        ctx.rest_relu = int(rest_relu_channels)
        if not 0 <= ctx.rest_relu <= Cr:
            raise ValueError("gru_step: rest_relu_channels %d outside [0, %d]" % (ctx.rest_relu, Cr))
        ctx.save_for_backward(*saved, *((rest,) if ctx.rest_relu else ()))
        ctx.packs, ctx.dims = packs, (B, C, Cr, H, W)
        return h

    @staticmethod
    def backward(ctx, g):
        if any(ctx.needs_input_grad[i] for i in (2, 4, 6, 8)):
            raise RuntimeError("gru_step is the frozen-weight path: no weight gradient")
        B, C, Cr, H, W = ctx.dims
        n = C * H * W
        new = lambda c: torch.empty((B, c, H, W), device=g.device, dtype=torch.float32)  # noqa: E731
        g = g.contiguous()
        d_rest = new(Cr)
        grads_p = [None, None, None, None]  # p_zr1, p_q1, p_zr2, p_q2
        if C % 32 == 0 and _GRU_EPILOGUES:
            # Elementwise backward kernels ride in the epilogues of the data-gradient convolutions: only the update
            # backward of the LAST half-step (its gradient arrives from outside) is a launch of its own.
            z1, r1, q1, h1 = ctx.saved_tensors[4:8]
            z0, r0, q0, h0 = ctx.saved_tensors[0:4]
            (b_zr1, b_q1, v1), (b_zr0, b_q0, v0) = ctx.packs[1], ctx.packs[0]
            dz1, dqc1, dh1, dzr1 = new(C), new(C), new(C), new(2 * C)
            _call("pcfa_gru_update_bwd", _ptr(z1), _ptr(q1), _ptr(h1), _ptr(g), _ptr(dz1), _ptr(dqc1), _ptr(dh1), z1.numel())
            _call("pcfa_sepconv5_gru_gates_bwd", _ptr(dqc1), C, Cr, _ptr(b_q1), _ptr(z1), _ptr(r1), _ptr(h1), _ptr(dz1),
                  _ptr(dh1), _ptr(dzr1), _ptr(dh1), _ptr(d_rest), 0, B, H, W, v1)
            dz0, dqc0, dh0, dzr0 = new(C), new(C), new(C), new(2 * C)
            _call("pcfa_sepconv5_gru_update_bwd", _ptr(dzr1), C, Cr, _ptr(b_zr1), _ptr(dh1), _ptr(z0), _ptr(q0), _ptr(h0),
                  _ptr(dz0), _ptr(dqc0), _ptr(dh0), _ptr(d_rest), B, H, W, v1)
            _call("pcfa_sepconv5_gru_gates_bwd", _ptr(dqc0), C, Cr, _ptr(b_q0), _ptr(z0), _ptr(r0), _ptr(h0), _ptr(dz0),
                  _ptr(dh0), _ptr(dzr0), _ptr(dh0), _ptr(d_rest), 1, B, H, W, v0)
            if ctx.rest_relu:
                _call("pcfa_sepconv5_fwd_split_masked", _ptr(dzr0), 2 * C, None, 0, _ptr(b_zr0), _ptr(dh0), C, 1,
                      _ptr(d_rest), 1, _ptr(ctx.saved_tensors[8]), ctx.rest_relu, B, C + Cr, H, W, v0)
            else:
                _call("pcfa_sepconv5_fwd_split", _ptr(dzr0), 2 * C, None, 0, _ptr(b_zr0), _ptr(dh0), C, 1, _ptr(d_rest), 1,
                      B, C + Cr, H, W, v0)
            return dh0, d_rest, None, dzr0, None, dqc0, None, dzr1, None, dqc1, None
        rest_started = 0
        for half in (1, 0):
            z, r, q, h = ctx.saved_tensors[4 * half: 4 * half + 4]
            b_zr, b_q, vertical = ctx.packs[half]
            dz, dqc, dh, drh, dzr = new(C), new(C), new(C), new(C), new(2 * C)
            _call("pcfa_gru_update_bwd", _ptr(z), _ptr(q), _ptr(h), _ptr(g), _ptr(dz), _ptr(dqc), _ptr(dh), z.numel())
            # d[rh | rest] of the q convolution: rh part fresh, rest part into the step's running sum
            _call("pcfa_sepconv5_fwd_split", _ptr(dqc), C, None, 0, _ptr(b_q), _ptr(drh), C, 0, _ptr(d_rest),
                  rest_started, B, C + Cr, H, W, vertical)
            rest_started = 1
            for b in range(B):
                o, oz = b * n, b * 2 * n
                _call("pcfa_gru_gates_bwd_acc", _ptr_off(z, o), _ptr_off(r, o), _ptr_off(h, o), _ptr_off(dz, o),
                      _ptr_off(drh, o), _ptr_off(dh, o), _ptr_off(dzr, oz), _ptr_off(dzr, oz + n), _ptr_off(dh, o), n)
            # d[h | rest] of the stacked z|r convolution: both parts accumulate; the step's last write of d_rest also
            # applies the deferred ReLU mask of the layer that produced `rest`
            if half == 0 and ctx.rest_relu:
                _call("pcfa_sepconv5_fwd_split_masked", _ptr(dzr), 2 * C, None, 0, _ptr(b_zr), _ptr(dh), C, 1,
                      _ptr(d_rest), 1, _ptr(ctx.saved_tensors[8]), ctx.rest_relu, B, C + Cr, H, W, vertical)
            else:
                _call("pcfa_sepconv5_fwd_split", _ptr(dzr), 2 * C, None, 0, _ptr(b_zr), _ptr(dh), C, 1, _ptr(d_rest), 1,
                      B, C + Cr, H, W, vertical)
            grads_p[2 * half], grads_p[2 * half + 1] = dzr, dqc
            g = dh
        return g, d_rest, None, grads_p[0], None, grads_p[1], None, grads_p[2], None, grads_p[3], None


class _Fanout(torch.autograd.Function):
    """x -> n aliases of x, one per consumer.  Forward moves no data; backward receives all n gradients at once and
    adds them with ONE launch (pcfa_sum_n) instead of the n-1 pairwise accumulations autograd performs when the same
    tensor feeds n nodes.  Consumers that contributed nothing are skipped."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [g.contiguous() for g in grads if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        _dev(*live)
        out = torch.empty_like(live[0])
        for i in range(0, len(live), 15):      # 16 pointers per launch: the running sum + 15 more
            part = ([out] if i else []) + live[i:i + 15]
            arr = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
            _call("pcfa_sum_n", arr, len(part), _ptr(out), out.numel())
        return out, None


class _ConvexUpsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, mask):
        _dev(flow, mask)
        N, C, H, W = flow.shape
        if C != 2 or tuple(mask.shape) != (N, 576, H, W):
            raise ValueError("convex_upsample: flow %s / mask %s (expected [N,2,H,W] and [N,576,H,W])"
                             % (tuple(flow.shape), tuple(mask.shape)))
        flow, mask = flow.contiguous(), mask.contiguous()
        out = torch.empty((N, 2, 8 * H, 8 * W), device=flow.device, dtype=torch.float32)
        _call("pcfa_convex_upsample_fwd", _ptr(flow), _ptr(mask), _ptr(out), N, H, W)
        ctx.save_for_backward(flow, mask)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        flow, mask = ctx.saved_tensors
        N, _, H, W = flow.shape
        g = g.contiguous()
        gflow, gmask = torch.empty_like(flow), torch.empty_like(mask)
        ws = torch.empty(int(_hip.load().pcfa_convex_upsample_workspace_floats(N, H, W)), device=g.device,
                         dtype=torch.float32)
        _call("pcfa_convex_upsample_bwd", _ptr(flow), _ptr(mask), _ptr(g), _ptr(gflow), _ptr(gmask), _ptr(ws), N, H, W)
        return gflow, gmask


def convex_upsample(flow, mask):
    """[N,2,H,W] -> [N,2,8H,8W] by the softmax-weighted 3x3 combination of RAFT.upsample_flow (raft.py:72-83): one
    streaming launch per direction instead of softmax + unfold + multiply + reduce + permute over 26 MB temporaries."""
    return _ConvexUpsample.apply(flow, mask)


class _FlowStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords1, delta, coords0):
        _dev(coords1, delta, coords0)
        if not (coords1.shape == delta.shape == coords0.shape):
            raise ValueError("flow_step: shapes differ: %s %s %s" % (tuple(coords1.shape), tuple(delta.shape),
                                                                     tuple(coords0.shape)))
        c1, d, c0 = coords1.contiguous(), delta.contiguous(), coords0.contiguous()
        c1n, fl = torch.empty_like(c1), torch.empty_like(c1)
        _call("pcfa_flow_step", _ptr(c1), _ptr(d), _ptr(c0), _ptr(c1n), _ptr(fl), c1.numel())
        ctx.set_materialize_grads(False)
        return c1n, fl

    @staticmethod
    def backward(ctx, g1, g2):
        g = g1 if g2 is None else g2 if g1 is None else g1 + g2
        return (g if ctx.needs_input_grad[0] else None, g if ctx.needs_input_grad[1] else None,
                (None if g is None else -g) if ctx.needs_input_grad[2] else None)


def flow_step(coords1, delta, coords0):
    """(coords1 + delta, coords1 + delta - coords0): the coordinate update of a refinement iteration and the flow the
    next iteration / the upsampler reads (models/raft/raft.py:122-137), one launch."""
    return _FlowStep.apply(coords1, delta, coords0)


def fanout(x, n):
    """n aliases of x whose gradients are summed by one kernel (see _Fanout)."""
    return _Fanout.apply(x, n) if n > 1 else (x,)


def gru_step(h, rest, halves, rest_relu_channels=0):
    """SepConvGRU update from precomputed context parts: halves = ((w_zr, p_zr, w_q, p_q) for the 1x5 half-step,
    (..) for the 5x1 half-step); see _GruStep.  rest_relu_channels = n > 0: rest[:, :n] are ReLU outputs whose producer
    ran with grad_premasked=True and has no other consumer -- the gradient returned for them is already multiplied
    by [rest > 0] (applied by the kernel that writes it last)."""
    (a, b, c, d), (e, f, g_, i_) = halves
    return _GruStep.apply(h, rest, a, b, c, d, e, f, g_, i_, int(rest_relu_channels))


def sepconv5(a, b, weight):
    """conv2d(cat([a, b], 1), weight, bias=None, padding='same') for a frozen (1,5) or (5,1) `weight`
    (SepConvGRU gate convolutions, models/raft/update.py:36-60); `b` may be None."""
    return _SepConv5.apply(a, b, weight)


def gru_gates_packed(zr, h, bias_zr=None, add_zr=None):
    """(z, r*h) from the stacked gate pre-activations zr (+ add_zr) = conv_{[Wz;Wr]}(.) of shape [B, 2C, H, W]."""
    return _GruGatesPacked.apply(zr, h, bias_zr, add_zr)


def gru_gates(zc, rc, h, bias_z=None, bias_r=None, add_z=None, add_r=None):
    """(z, r*h) with z = sigmoid(zc + add_z + bias_z), r = sigmoid(rc + add_r + bias_r); biases are frozen."""
    return _GruGates.apply(zc, rc, h, bias_z, bias_r, add_z, add_r)


def gru_update(z, qc, h, bias_q=None, add_q=None):
    """(1 - z) * h + z * tanh(qc + add_q + bias_q)."""
    return _GruUpdate.apply(z, qc, h, bias_q, add_q)


def bias_relu(x, bias=None):
    """relu(x + bias[None, :, None, None]) for a frozen bias (conv -> +bias -> ReLU in one pass)."""
    return _BiasRelu.apply(x, bias)
