"""Autograd-aware operators of the PCFA hot path, backed by libpcfa_hip.so.

This is the ONLY operator implementation inside the package.  Every op checks
that its tensors live on a HIP device and raises otherwise (no CPU fallback);
launches go to torch's current stream through the C-ABI, so they order with
the surrounding MIOpen/hipBLASLt work and can be captured into a hipGraph.

Operator boundaries mirrored (reference file:line):
  CorrBlock                      models/raft/corr.py:12-60 (== models/gma/corr.py:15-63)
  spatial_correlation_sample     .../spatial_correlation_sampler/spatial_correlation_sampler.py:9-91
  flownet_correlation, resample2d, channelnorm   models/FlowNet/{correlation,resample2d,channelnorm}_package/*.py
  pwc_warp, dense_block          models/PWCNet/PWCNet.py:166-206, :234-323
  conv3x3, conv3x3_fewout, conv3x3_cat, conv_fewin, sepconv5, gru_step, bias_relu, flow_step, convex_upsample
                                 models/raft/update.py, models/raft/extractor.py, PWCNet.py:29-38, FlowNet/submodules.py
  instance_norm_relu, add_relu   models/raft/extractor.py:23-58
  box_transform                  helper_functions/own_models.py:62-85
  extract_deltas(_joint)         attack_PCFA.py:20-37
  loss_delta_constraint, avg_epe, two_norm_*   helper_functions/losses.py
"""
import ctypes
import os
import weakref

import torch

from . import _hip
from .lbfgs import LBFGS  # noqa: F401  (the attack loop's optimiser: torch.optim.LBFGS semantics, HIP vector math)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "pcfa_amd HIP operator called with a %s tensor: the MI355X path has no CPU fallback"
                % t.device)
        if t.dtype != torch.float32:
            raise TypeError("pcfa_amd kernels compute in float32, got %s" % t.dtype)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _ptr_off(t, offset_floats):
    return ctypes.c_void_p(0 if t is None else t.data_ptr() + 4 * offset_floats)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class LaunchProfiler:
    """Optional per-launch timing: HIP events recorded on the launch stream around every C-ABI call.
    Used by bench.py to measure kernel durations live inside the timed region (off by default)."""

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.events = {}

    def wants(self, name):
        return self.names is None or name in self.names

    def summary(self):
        """name -> (mean microseconds, launches); synchronises the device."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.events.items():
            tot = sum(s.elapsed_time(e) for s, e in pairs)
            out[name] = (1e3 * tot / len(pairs), len(pairs))
        return out


class DispatchTimer:
    """Kernel durations from hipEvents attached to the dispatch packet itself (pcfa_timing_arm ->
    hipExtLaunchKernel).  Unlike an event bracket around a launch, which inserts two barrier packets
    (measured: +4..7 us per launch on MI355X), these events carry the packet's own begin/end timestamps --
    the same source rocprofv3's kernel trace reads.  Used by bench.py for the roofline figures.

    `plan` maps a C-ABI entry point to [(label, nth kernel it launches)]; see include/pcfa_hip.h for the
    launch order of the multi-kernel entry points."""

    DEFAULT_PLAN = {
        "pcfa_corr_lookup_fwd": [("corr_lookup_fwd", 0)],
        "pcfa_corr_lookup_bwd": [("corr_lookup_bwd", 0)],
        "pcfa_corr_pyramid_fwd": [("corr_pyramid_gemm_fwd", 0)],
        "pcfa_corr_pyramid_bwd": [("corr_pyramid_gemm_dfmap1", 0), ("corr_pyramid_gemm_df2ext", 2)],
        "pcfa_corr_pyramid_bwd_windows": [("corr_pyramid_gemm_dfmap1", 2), ("corr_pyramid_gemm_df2ext", 4)],
        "pcfa_corr_f2ext_fwd": [("corr_f2ext_fwd", 0)],
        "pcfa_spatial_corr_fwd": [("spatial_corr_fwd", 0)],
        "pcfa_spatial_corr_bwd": [("spatial_corr_bwd_in1", 0), ("spatial_corr_bwd_in2", 1)],
        "pcfa_flownet_corr_fwd": [("flownet_corr_fwd", 0)],
        "pcfa_flownet_corr_bwd": [("flownet_corr_bwd_in1", 0), ("flownet_corr_bwd_in2", 1)],
        "pcfa_resample2d_fwd": [("resample2d_fwd", 0)],
        "pcfa_resample2d_bwd": [("resample2d_bwd", 1)],  # kernel 0 clears grad_in1
        "pcfa_channelnorm_fwd": [("channelnorm_fwd", 0)],
        "pcfa_channelnorm_bwd": [("channelnorm_bwd", 0)],
        "pcfa_box_transform_fwd": [("box_transform_fwd", 0)],
        "pcfa_box_transform_bwd": [("box_transform_bwd", 0)],
        "pcfa_flow_loss_fwd": [("flow_loss_partial", 0)],
        "pcfa_gru_gates_fwd": [("gru_gates_fwd", 0)],
        "pcfa_gru_gates_bwd": [("gru_gates_bwd", 0)],
        "pcfa_gru_update_fwd": [("gru_update_fwd", 0)],
        "pcfa_gru_update_bwd": [("gru_update_bwd", 0)],
        "pcfa_conv_fewin_fwd": [("conv_fewin_fwd", 0)],
        "pcfa_pwc_warp_fwd": [("pwc_warp_fwd", 0)],
        "pcfa_pwc_warp_bwd": [("pwc_warp_bwd", 1)],
        "pcfa_pwc_warp_bwd_det": [("pwc_warp_bwd", 1)],
        "pcfa_conv3x3_fewout_fwd": [("conv3x3_fewout_fwd", 0)],
        "pcfa_conv3x3_fewout_bwd": [("conv3x3_fewout_bwd", 0)],
        "pcfa_instnorm_fwd": [("instnorm_stats_fwd", 0), ("instnorm_apply_fwd", 1)],
        "pcfa_instnorm_bwd": [("instnorm_stats_bwd", 0), ("instnorm_apply_bwd", 1)],
        "pcfa_add_relu_fwd": [("add_relu_fwd", 0)],
        "pcfa_bias_relu_fwd": [("bias_relu_fwd", 0)],
        "pcfa_relu_bwd": [("relu_bwd", 0)],
    }

    EVENT_FLAGS = 0x20000000  # hipEventDisableSystemFence

    def __init__(self, plan=None):
        self.plan = dict(self.DEFAULT_PLAN if plan is None else plan)
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pairs = {}

    def new_pair(self, name):
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        for e in (e0, e1):
            # timing-only events: without hipEventDisableSystemFence the dispatch they ride on ends with a
            # SYSTEM-scope release (write-back of every dirty L2 line, also those of earlier kernels), which a
            # plain or graph-replayed launch does not pay -- rocprofv3 shows the same kernel 1.7 us longer with
            # default events attached (tools/dev/lookup_trace_split.py)
            err = self.hip.hipEventCreateWithFlags(ctypes.byref(e), self.EVENT_FLAGS)
            if err != 0:
                raise RuntimeError("hipEventCreateWithFlags failed: %d" % err)
        self.pairs.setdefault(name, []).append((e0, e1))
        return e0, e1

    def summary(self):
        """name -> (mean microseconds, launches); synchronises the device and releases the events."""
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.pairs.items():
            tot, n = 0.0, 0
            for e0, e1 in pairs:
                ms = ctypes.c_float()
                if self.hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1) == 0:  # else: never launched
                    tot += ms.value
                    n += 1
                self.hip.hipEventDestroy(e0)
                self.hip.hipEventDestroy(e1)
            if n:
                out[name] = (1e3 * tot / n, n)
        self.pairs = {}
        return out


_profiler = None
_dispatch_timer = None


def set_launch_profiler(profiler):
    global _profiler
    _profiler = profiler


def set_dispatch_timer(timer):
    global _dispatch_timer
    _dispatch_timer = timer


# ---- work accounting for the roofline rows of bench.py (off unless a recorder is set) -----------------------------------
# family -> [direct-form flop (or algorithmic bytes), flop the matrix cores actually issue, calls]; the arithmetic of a
# Winograd kernel is its direct-form flop divided by the transform's saving: F(2x2,3x3) 36 / 16, F(4x4,3x3) 144 / 36,
# F(2,5) 10 / 6.
_work = None


def set_work_recorder(rec):
    """rec: a dict that _call fills per kernel family while set (None: off)."""
    global _work
    _work = rec


def _note_work(family, direct, issued):
    e = _work.setdefault(family, [0.0, 0.0, 0])
    e[0] += direct
    e[1] += issued
    e[2] += 1


def _conv3x3_work(B, K, N, H, W):
    direct = 2.0 * 9 * K * N * B * H * W
    if _hip.load().pcfa_conv3x3_algo(B, K, N, H, W) == 43:
        _note_work("conv3x3_f43", direct, direct / 4.0)
    else:
        _note_work("conv3x3_winograd", direct, direct / 2.25)


def _sepconv5_work(B, Ca, Cb, Cout, H, W, vertical):
    direct = 2.0 * 5 * (Ca + Cb) * Cout * B * H * W
    if _hip.load().pcfa_sepconv5_uses_winograd(B, Ca, Cb, Cout, H, W, int(vertical)):
        _note_work("sepconv5_winograd", direct, direct * 0.6)
    else:
        _note_work("sepconv5_direct", direct, direct)


_WORK_TABLE = {   # entry point -> accounting of its positional arguments (the order of include/pcfa_hip.h)
    "pcfa_conv3x3_run": lambda a: _conv3x3_work(a[6], a[7], a[8], a[9], a[10]),
    "pcfa_conv3x3_act_fwd_pair": lambda a: (_conv3x3_work(1, a[4], a[5], a[12], a[13]),
                                            _conv3x3_work(1, a[10], a[11], a[12], a[13])),
    "pcfa_sepconv5_fwd": lambda a: _sepconv5_work(a[6], a[1], a[3], a[7], a[8], a[9], a[10]),
    "pcfa_sepconv5_fwd_split": lambda a: _sepconv5_work(a[10], a[1], a[3], a[11], a[12], a[13], a[14]),
    "pcfa_sepconv5_fwd_split_masked": lambda a: _sepconv5_work(a[12], a[1], a[3], a[13], a[14], a[15], a[16]),
    "pcfa_sepconv5_gru_gates_fwd": lambda a: _sepconv5_work(a[9], a[1], a[3], 2 * a[1], a[10], a[11], a[12]),
    "pcfa_sepconv5_gru_update_fwd": lambda a: _sepconv5_work(a[10], a[1], a[3], a[1], a[11], a[12], a[13]),
    "pcfa_sepconv5_gru_gates_bwd": lambda a: _sepconv5_work(a[13], a[1], 0, a[1] + a[2], a[14], a[15], a[16]),
    "pcfa_sepconv5_gru_update_bwd": lambda a: _sepconv5_work(a[12], 2 * a[1], 0, a[1] + a[2], a[13], a[14], a[15]),
    # instance norm: algorithmic traffic = x in + y out (forward), x + grad_out in + grad_x out (backward)
    "pcfa_instnorm_fwd": lambda a: _note_work("instnorm_fwd", 2.0 * a[4] * a[5] * 4, 0.0),
    "pcfa_instnorm_bwd": lambda a: _note_work("instnorm_bwd", 3.0 * a[5] * a[6] * 4, 0.0),
    # streams: input once + the small output (flow-prediction convolutions), elementwise passes
    "pcfa_conv3x3_fewout_fwd": lambda a: _note_work("conv3x3_fewout_fwd", 4.0 * a[5] * (a[6] + a[7]) * a[8] * a[9], 0.0),
    "pcfa_conv3x3_fewout_bwd": lambda a: _note_work("conv3x3_fewout_bwd", 4.0 * a[3] * (a[4] + a[5]) * a[6] * a[7], 0.0),
    "pcfa_relu_bwd": lambda a: _note_work("relu_bwd", 12.0 * a[3], 0.0),
    "pcfa_relu_bwd2": lambda a: _note_work("relu_bwd2", 20.0 * a[5], 0.0),
    "pcfa_add_relu_fwd": lambda a: _note_work("add_relu_fwd", 12.0 * a[3], 0.0),
    "pcfa_conv_fewin_packed_fwd": lambda a: _note_work("conv_fewin_fwd", *(2 * [2.0 * a[5] * a[9] * a[9] * a[6] * a[4] * a[7] * a[8]])),
}


def _call(name, *args):
    """Invoke C-ABI entry point `name` on torch's current stream and raise on a non-zero status."""
    fn = getattr(_hip.load(), name)
    if _work is not None and name in _WORK_TABLE:
        _WORK_TABLE[name](args)
    prof = _profiler
    timer = _dispatch_timer
    if timer is not None and name in timer.plan:
        lib = _hip.load()
        for label, nth in timer.plan[name]:
            e0, e1 = timer.new_pair(label)
            _hip.check(lib.pcfa_timing_arm(e0, e1, nth), "pcfa_timing_arm")
        try:
            status = fn(*args, _stream())
        finally:
            lib.pcfa_timing_arm(None, None, -1)  # drop pairs the entry point did not reach
    elif prof is not None and prof.wants(name):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        status = fn(*args, _stream())
        e.record()
        prof.events.setdefault(name, []).append((s, e))
    else:
        status = fn(*args, _stream())
    _hip.check(status, name)


# --------------------------------------------------------------------------- #
# RAFT / GMA correlation pyramid
# --------------------------------------------------------------------------- #
PYRAMID_BWD_WINDOWS = True   # False: the dense backward products (A/B in tools, parity tests)


class _CorrState:
    """Device buffers shared by the build node and its lookup nodes."""
    __slots__ = ("B", "D", "H", "W", "L", "r", "slab", "pyr", "f2ext", "dpyr", "token_grad", "coords_bwd")


class _CorrBuild(torch.autograd.Function):
    """fmap1, fmap2 -> 1-element token; the pyramid itself lives in `state`.

    The token only carries the autograd dependency: every lookup consumes it, so
    this node's backward runs after ALL lookup backwards have accumulated into
    state.dpyr, and performs the two GEMMs of the volume's backward once.
    """

    @staticmethod
    def forward(ctx, fmap1, fmap2, state):
        lib = _hip.load()
        B, D, H, W = fmap1.shape
        f1 = fmap1.contiguous()
        f2 = fmap2.contiguous()
        slab = state.slab
        state.f2ext = torch.empty((B, D, slab), device=f1.device, dtype=torch.float32)
        state.pyr = torch.empty((B * H * W, slab), device=f1.device, dtype=torch.float32)
        _call("pcfa_corr_f2ext_fwd", _ptr(f2), _ptr(state.f2ext), B, D, H, W, state.L)
        _call("pcfa_corr_pyramid_fwd", _ptr(f1), _ptr(state.f2ext), _ptr(state.pyr), B, D, H, W,
                                             state.L)
        ctx.state = state
        ctx.save_for_backward(f1)
        return torch.zeros(1, device=f1.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, grad_token):
        st = ctx.state
        (f1,) = ctx.saved_tensors
        if st.dpyr is None:  # no lookup contributed a gradient
            z = torch.zeros_like(f1)
            st.token_grad = None
            return z, z.clone(), None
        lib = _hip.load()
        B, D, H, W = st.B, st.D, st.H, st.W
        df1 = torch.empty_like(f1)
        df2 = torch.empty_like(f1)
        # the coordinates of every lookup that accumulated into dpyr: the products skip what no window touched
        # (the per-block segment record of corr_window_segments_kernel holds four levels: more levels -> dense products)
        cs = st.coords_bwd if (st.coords_bwd and len(st.coords_bwd) <= 32 and PYRAMID_BWD_WINDOWS and st.L <= 4) else []
        if cs:
            nbytes = lib.pcfa_corr_pyramid_bwd_windows_workspace_bytes(B, D, H, W, st.L)
            ws = torch.empty((nbytes + 3) // 4, device=f1.device, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * len(cs))(*[c.data_ptr() for c in cs])
            _call("pcfa_corr_pyramid_bwd_windows", _ptr(st.dpyr), _ptr(f1), _ptr(st.f2ext), _ptr(df1), _ptr(df2), _ptr(ws),
                  ctypes.c_size_t(nbytes), ptrs, len(cs), st.r, B, D, H, W, st.L)
            if _work is not None:
                # executed matrix work of the two sparse products: the K segments the kernels walked, read back from the
                # workspace (csrc/corr_pyramid.hip: per 128-wide column block {count, (begin, end) x 4, pad} ints behind
                # the split-K area; segA = blocks of dfmap1's Q columns, segB = blocks of df2ext's slab columns)
                base = (int(lib.pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, st.L)) + 15) & ~15
                nbA, nbB = -(-(H * W) // 128), -(-st.slab // 128)
                seg = ws.view(torch.int32)[base // 4: base // 4 + 10 * B * (nbA + nbB)].cpu().view(-1, 10).long()
                live = torch.arange(4)[None, :] < seg[:, :1]
                k = ((seg[:, 2:9:2] - seg[:, 1:8:2]) * live).sum(1)
                dense = 2.0 * B * D * (H * W) ** 2
                _note_work("corr_pyramid_gemm_dfmap1", dense, 2.0 * D * 128 * float(k[:B * nbA].sum()))
                _note_work("corr_pyramid_gemm_df2ext", dense, 2.0 * D * 128 * float(k[B * nbA:].sum()))
        else:   # the dense products under their own entry point (and their own launch indices in DispatchTimer's plan)
            nbytes = lib.pcfa_corr_pyramid_bwd_workspace_bytes(B, D, H, W, st.L)
            ws = torch.empty((nbytes + 3) // 4, device=f1.device, dtype=torch.float32)
            _call("pcfa_corr_pyramid_bwd", _ptr(st.dpyr), _ptr(f1), _ptr(st.f2ext), _ptr(df1), _ptr(df2), _ptr(ws),
                  ctypes.c_size_t(nbytes), B, D, H, W, st.L)
        st.dpyr = None
        st.token_grad = None
        st.coords_bwd = None
        return df1, df2, None


def _token_grad(st, device):
    """The 1-element token only orders the build node behind every lookup node: ONE lookup per backward pass hands it a
    (zero) gradient, the others return None -- twelve zeros(1) fills and eleven 1-element accumulations per closure
    otherwise (each a kernel launch)."""
    if st.token_grad is None:
        st.token_grad = torch.zeros(1, device=device, dtype=torch.float32)
        return st.token_grad
    return None


class _CorrLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, token, coords, state):
        if coords.requires_grad and torch.is_grad_enabled():
            # models/raft/corr.py's bilinear_sampler differentiates w.r.t. coords; RAFT / GMA detach them
            # (raft.py:122-123) and this operator does not implement that gradient: refuse instead of returning zeros
            raise RuntimeError("CorrBlock lookup: coords.requires_grad is not supported (detach the coordinates, as "
                               "models/raft/raft.py:122-123 does)")
        lib = _hip.load()
        st = state
        c = coords.contiguous()
        n1 = 2 * st.r + 1
        out = torch.empty((st.B, st.L * n1 * n1, st.H, st.W), device=c.device, dtype=torch.float32)
        _call("pcfa_corr_lookup_fwd", _ptr(st.pyr), _ptr(c), _ptr(out), st.B, st.H, st.W, st.L, st.r)
        ctx.state = st
        ctx.save_for_backward(c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        st = ctx.state
        (c,) = ctx.saved_tensors
        lib = _hip.load()
        if st.dpyr is None:
            st.dpyr = torch.zeros_like(st.pyr)
            st.coords_bwd = []
        st.coords_bwd.append(c)
        g = grad_out.contiguous()
        _call("pcfa_corr_lookup_bwd", _ptr(st.dpyr), _ptr(c), _ptr(g), st.B, st.H, st.W, st.L, st.r)
        return _token_grad(st, g.device), None, None


_convc1_packs = {}


def _convc1_packed(weight):
    """pcfa_lookup_convc1_pack_weights of a frozen [256, 324, 1, 1] weight (both operand orders), cached per version."""
    key = id(weight)
    hit = _convc1_packs.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != weight._version:
        lib = _hip.load()
        cout = weight.shape[0]
        w = weight.detach().reshape(cout, -1).contiguous()
        packed = torch.empty(int(lib.pcfa_lookup_convc1_packed_floats(cout)), device=w.device, dtype=torch.float32)
        _call("pcfa_lookup_convc1_pack_weights", _ptr(w), _ptr(packed), cout, w.shape[1])
        hit = (weakref.ref(weight, lambda _r, k=key: _convc1_packs.pop(k, None)), weight._version, packed)
        _convc1_packs[key] = hit
    return hit[2]


class _CorrLookupConv(torch.autograd.Function):
    """relu(convc1(lookup(coords))) in one launch per direction (pcfa_lookup_convc1_fwd / _bwd): the lookup node of
    _CorrLookup with the motion encoder's 1x1 convolution (frozen weight) folded in.  Backward accumulates into the
    shared state.dpyr exactly like _CorrLookup."""

    @staticmethod
    def forward(ctx, token, coords, state, weight, bias, relu):
        if coords.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("CorrBlock lookup: coords.requires_grad is not supported (detach the coordinates, as "
                               "models/raft/raft.py:122-123 does)")
        st = state
        c = coords.contiguous()
        packed = _convc1_packed(weight)
        out = torch.empty((st.B, weight.shape[0], st.H, st.W), device=c.device, dtype=torch.float32)
        _call("pcfa_lookup_convc1_fwd", _ptr(st.pyr), _ptr(c), _ptr(packed), _ptr(bias), _ptr(out), st.B, st.H, st.W,
              st.L, st.r, weight.shape[0], int(relu))
        ctx.state, ctx.packed, ctx.relu, ctx.cout = st, packed, int(relu), weight.shape[0]
        ctx.save_for_backward(c, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            raise RuntimeError("lookup_conv is the frozen-weight path: no weight / bias gradient")
        st = ctx.state
        c, out = ctx.saved_tensors
        if st.dpyr is None:
            st.dpyr = torch.zeros_like(st.pyr)
            st.coords_bwd = []
        st.coords_bwd.append(c)
        g = grad_out.contiguous()
        _call("pcfa_lookup_convc1_bwd", _ptr(st.dpyr), _ptr(c), _ptr(ctx.packed), _ptr(out), _ptr(g), st.B, st.H, st.W,
              st.L, st.r, ctx.cout, ctx.relu)
        return _token_grad(st, g.device), None, None, None, None, None


class CorrBlock:
    """Drop-in for models/raft/corr.py:12-50 -- same constructor and __call__."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        _dev(fmap1, fmap2)
        if fmap1.shape != fmap2.shape or fmap1.dim() != 4:
            raise ValueError("CorrBlock expects two [B,D,H,W] feature maps of equal shape")
        lib = _hip.load()
        self.num_levels = num_levels
        self.radius = radius
        st = _CorrState()
        st.B, st.D, st.H, st.W = fmap1.shape
        st.L, st.r = num_levels, radius
        st.slab = lib.pcfa_corr_slab_floats(st.H, st.W, num_levels)
        if st.slab <= 0 or (st.H >> (num_levels - 1)) < 1 or (st.W >> (num_levels - 1)) < 1:
            raise ValueError("feature map %dx%d too small for %d pyramid levels" % (st.H, st.W, num_levels))
        st.dpyr = None
        st.token_grad = None
        st.coords_bwd = None
        self._state = st
        self._token = _CorrBuild.apply(fmap1, fmap2, st)

    def __call__(self, coords):
        _dev(coords)
        return _CorrLookup.apply(self._token, coords, self._state)

    def lookup_conv_relu(self, coords, weight, bias, relu=True):
        """relu(conv1x1(self(coords), weight, bias)) without materialising the lookup (update.py:79-93 convc1);
        None when the shape is not the fused kernel's (4 levels, radius 4, 256 x 324 weight, bias present)."""
        if (self.num_levels != 4 or self.radius != 4 or bias is None or weight.dim() != 4
                or tuple(weight.shape) != (256, 324, 1, 1) or weight.requires_grad or bias.requires_grad):
            return None
        _dev(coords, weight, bias)
        return _CorrLookupConv.apply(self._token, coords, self._state, weight, bias, relu)

    @property
    def corr_pyramid(self):
        """Per-level tensors [B*Q,1,H_l,W_l] gathered out of the tiled slab matrix (for inspection/tests)."""
        st = self._state
        return [st.pyr[:, idx.to(st.pyr.device)].reshape(-1, 1, h, w)
                for (idx, h, w) in tiled_index_maps(st.H, st.W, st.L)]


def tiled_index_maps(H, W, num_levels):
    """[(index tensor [H_l*W_l] into a query slab, H_l, W_l)] -- the 4x4-tile layout of include/pcfa_hip.h."""
    out, off, h, w = [], 0, H, W
    for _ in range(num_levels):
        tw = (w + 3) // 4
        ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        idx = off + ((ys // 4) * tw + xs // 4) * 16 + (ys % 4) * 4 + xs % 4
        out.append((idx.reshape(-1), h, w))
        off += ((h + 3) // 4) * tw * 16
        h, w = h // 2, w // 2
    return out


# --------------------------------------------------------------------------- #
# PWC-Net cost volume
# --------------------------------------------------------------------------- #
class SpatialCorrelationSamplerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                dilation_patch=1):
        _dev(input1, input2)
        # the reference's CPU build reads through accessors (any strides); the kernels want dense NCHW
        input1, input2 = input1.contiguous(), input2.contiguous()
        lib = _hip.load()
        kH, kW = _pair(kernel_size)
        pH, pW = _pair(patch_size)
        padH, padW = _pair(padding)
        dilH, dilW = _pair(dilation)
        dpH, dpW = _pair(dilation_patch)
        dH, dW = _pair(stride)
        B, C, iH, iW = input1.shape
        oH, oW = ctypes.c_int(), ctypes.c_int()
        _hip.check(lib.pcfa_spatial_corr_out_size(iH, iW, kH, kW, padH, padW, dilH, dilW, dH, dW,
                                                  ctypes.byref(oH), ctypes.byref(oW)), "pcfa_spatial_corr_out_size")
        out = torch.empty((B, pH, pW, oH.value, oW.value), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, iH, iW, kH, kW, pH, pW, padH, padW, dilH, dilW, dpH, dpW, dH, dW)
        _call("pcfa_spatial_corr_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        lib = _hip.load()
        g = grad_output.contiguous()
        g1 = torch.empty_like(input1)
        g2 = torch.empty_like(input2)
        _call("pcfa_spatial_corr_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2),
                                             *ctx.params)
        return g1, g2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride, padding,
                                                   dilation, dilation_patch)


class _PwcCostVolume(torch.autograd.Function):
    """leaky_relu(spatial_correlation_sample(a, b, patch 9) / C) -- PWCNet.py:45-58 + the LeakyReLU that follows every
    call (:249,264,278,292,308) -- as ONE forward launch (scale and activation in the epilogue) and ONE backward
    launch (mask * scale applied to the gradient taps while they are staged; both input gradients)."""

    @staticmethod
    def forward(ctx, input1, input2, slope):
        _dev(input1, input2)
        input1, input2 = input1.contiguous(), input2.contiguous()
        B, C, H, W = input1.shape
        out = torch.empty((B, 81, H, W), device=input1.device, dtype=torch.float32)
        ctx.args = (B, C, H, W, 1.0 / C, float(slope))
        _call("pcfa_cost_volume9_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.args)
        ctx.save_for_backward(input1, input2, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2, out = ctx.saved_tensors
        g = grad_output.contiguous()
        if g.data_ptr() % 16:          # a view with a storage offset: the kernel stages 16-B pieces
            g = g.clone()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_cost_volume9_bwd", _ptr(input1), _ptr(input2), _ptr(out), _ptr(g), _ptr(g1), _ptr(g2), *ctx.args)
        return g1, g2, None


def pwc_cost_volume(input1, input2, slope=0.1):
    """PWC-Net's `leakyRELU(correlate(input1, input2))`: [B,C,H,W] x2 -> [B,81,H,W]."""
    misaligned = any(t.is_contiguous() and t.data_ptr() % 16 for t in (input1, input2))
    if input1.shape[-1] % 4 != 0 or input1.shape != input2.shape or misaligned:   # the fused kernels stage 16-B pieces
        out = spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=9, stride=1)
        b, ph, pw, h, w = out.size()
        return torch.nn.functional.leaky_relu(out.view(b, ph * pw, h, w) / input1.size(1), slope)
    return _PwcCostVolume.apply(input1, input2, slope)


# --------------------------------------------------------------------------- #
# FlowNet2's native operators (models/FlowNet/{correlation,resample2d,channelnorm}_package)
# --------------------------------------------------------------------------- #
class CorrelationFunction(torch.autograd.Function):
    """correlation_package/correlation.py:10-51 on pcfa_flownet_corr_fwd/bwd (no rbot1/rbot2 scratch copies)."""

    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1):
        _dev(input1, input2)
        input1, input2 = input1.contiguous(), input2.contiguous()
        if input1.shape != input2.shape or input1.dim() != 4:
            raise RuntimeError("Correlation: inputs must be two [B,C,H,W] tensors of the same shape")
        lib = _hip.load()
        B, C, H, W = input1.shape
        ctx.params = (B, C, H, W, int(pad_size), int(kernel_size), int(max_displacement), int(stride1), int(stride2))
        oc, oH, oW = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _hip.check(lib.pcfa_flownet_corr_out_size(H, W, *ctx.params[4:], ctypes.byref(oc), ctypes.byref(oH),
                                                  ctypes.byref(oW)), "pcfa_flownet_corr_out_size")
        out = torch.empty((B, oc.value, oH.value, oW.value), device=input1.device, dtype=torch.float32)
        _call("pcfa_flownet_corr_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g = grad_output.contiguous()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_flownet_corr_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2), *ctx.params)
        return g1, g2, None, None, None, None, None, None


def flownet_correlation(input1, input2, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2,
                        corr_multiply=1):
    """Correlation.forward (correlation_package/correlation.py:53-67)."""
    return CorrelationFunction.apply(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                                     corr_multiply)


class Resample2dFunction(torch.autograd.Function):
    """resample2d_package/resample2d.py:12-43."""

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, bilinear=True):
        _dev(input1, input2)
        if not (input1.is_contiguous() and input2.is_contiguous()):
            raise AssertionError("Resample2d: inputs must be contiguous")  # the reference asserts (resample2d.py:16-17)
        B, C, iH, iW = input1.shape
        b, two, H, W = input2.shape
        if b != B or two != 2:
            raise RuntimeError("Resample2d: flow must be [B,2,H,W] with the batch size of input1")
        out = torch.empty((B, C, H, W), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, iH, iW, H, W, int(kernel_size), int(bool(bilinear)))
        _call("pcfa_resample2d_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g = grad_output.contiguous()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_resample2d_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2), *ctx.params)
        return g1, g2, None, None


def resample2d(input1, input2, kernel_size=1, bilinear=True):
    """Resample2d.forward (resample2d_package/resample2d.py:45-56)."""
    return Resample2dFunction.apply(input1.contiguous(), input2, kernel_size, bilinear)


class ChannelNormFunction(torch.autograd.Function):
    """channelnorm_package/channelnorm.py:11-36."""

    @staticmethod
    def forward(ctx, input1, norm_deg=2):
        _dev(input1)
        if not input1.is_contiguous():
            raise AssertionError("ChannelNorm: input must be contiguous")  # channelnorm.py:15
        B, C, H, W = input1.shape
        out = torch.empty((B, 1, H, W), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, H * W, int(norm_deg))
        _call("pcfa_channelnorm_fwd", _ptr(input1), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, out = ctx.saved_tensors
        g = grad_output.contiguous()
        g1 = torch.empty_like(input1)
        _call("pcfa_channelnorm_bwd", _ptr(input1), _ptr(out), _ptr(g), _ptr(g1), *ctx.params)
        return g1, None


def channelnorm(input1, norm_deg=2):
    """ChannelNorm.forward (channelnorm_package/channelnorm.py:38-45)."""
    return ChannelNormFunction.apply(input1, norm_deg)


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


WARP_BWD_DETERMINISTIC = True   # fixed-point scatter (bit-reproducible); False: hardware fp32 atomics


class _PwcWarp(torch.autograd.Function):
    """PWCDCNet.warp (models/PWCNet/PWCNet.py:166-206) on pcfa_pwc_warp_fwd/bwd."""

    @staticmethod
    def forward(ctx, x, flo, mask_threshold):
        _dev(x, flo)
        x, flo = x.contiguous(), flo.contiguous()
        B, C, H, W = x.shape
        if tuple(flo.shape) != (B, 2, H, W):
            raise ValueError("pwc_warp: flow %s does not match features %s" % (tuple(flo.shape), tuple(x.shape)))
        out = torch.empty_like(x)
        ctx.params = (B, C, H, W, float(mask_threshold))
        _call("pcfa_pwc_warp_fwd", _ptr(x), _ptr(flo), _ptr(out), *ctx.params)
        ctx.save_for_backward(x, flo)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, flo = ctx.saved_tensors
        g = g.contiguous()
        gx, gf = torch.empty_like(x), torch.empty_like(flo)
        if WARP_BWD_DETERMINISTIC:
            B, C, H, W, thr = ctx.params
            nws = int(_hip.load().pcfa_pwc_warp_bwd_det_workspace_bytes(B, C, H, W))
            ws = torch.empty((nws + 7) // 8, device=x.device, dtype=torch.int64)
            _call("pcfa_pwc_warp_bwd_det", _ptr(x), _ptr(flo), _ptr(g), _ptr(gx), _ptr(gf), _ptr(ws), nws, B, C, H, W,
                  thr)
        else:
            _call("pcfa_pwc_warp_bwd", _ptr(x), _ptr(flo), _ptr(g), _ptr(gx), _ptr(gf), *ctx.params)
        return gx, gf, None


def pwc_warp(x, flo, mask_threshold=0.0001):
    """Backward-warp x by flo with PWC-Net's validity mask: one launch forward, two backward."""
    return _PwcWarp.apply(x, flo, mask_threshold)


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


CONV_S2_BWD = True   # data gradient on pcfa_conv_s2_bwd where it applies (False: library gradient; tools/dev A/B)
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
    def forward(ctx, x, weight, bias, act, slope, grad_premasked=False):
        _dev(x, weight, bias)
        ctx.grad_premasked = bool(grad_premasked)   # the consumer applies this layer's activation backward (conv3x3)
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
        if CONV_S2_BWD and _hip.load().pcfa_conv_s2_bwd_supported(Cin, N, k, H, W):
            gx = torch.empty(ctx.xshape, device=g.device, dtype=torch.float32)
            _call("pcfa_conv_s2_bwd", _ptr(g), _ptr(_s2_bwd_packed(weight)), _ptr(gx), B, Cin, N, H, W, k)
        else:   # the stem's gradient and ragged widths: library
            gx = torch.nn.grad.conv2d_input(ctx.xshape, weight, g, stride=2, padding=k // 2)
        return gx, None, None, None, None, None


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


def conv_s2(x, weight, bias=None, relu=False, leaky_slope=None, grad_premasked=False):
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
    return _ConvS2.apply(x, weight, bias, act, 0.0 if leaky_slope is None else leaky_slope, bool(grad_premasked))


class _Conv3x3FewOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight, bias)
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
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("conv3x3_fewout is the frozen-weight path: no weight / bias gradient")
        (w,) = ctx.saved_tensors
        B, K, N, H, W = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_conv3x3_fewout_bwd", _ptr(g), _ptr(w), _ptr(gx), B, K, N, H, W)
        return gx, None, None


def conv3x3_fewout(x, weight, bias=None):
    """conv2d(x, weight, bias, stride=1, padding=1) for a frozen 3x3 weight with at most 4 output channels (the
    flow-prediction layers): a streaming kernel instead of a padded matrix-core tile."""
    return _Conv3x3FewOut.apply(x, weight, bias)


class _Deconv4s2FewOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight, bias)
        if weight.dim() != 4 or tuple(weight.shape[2:]) != (4, 4) or not 1 <= weight.shape[1] <= 4:
            raise ValueError("deconv4s2_fewout expects a [K, N<=4, 4, 4] ConvTranspose2d weight, got %s"
                             % (tuple(weight.shape),))
        x = x.contiguous()
        B, K, H, W = x.shape
        N = weight.shape[1]
        if weight.shape[0] != K:
            raise ValueError("deconv4s2_fewout: input %s does not match weight %s" % (tuple(x.shape), tuple(weight.shape)))
        w = weight.detach().contiguous()
        out = torch.empty((B, N, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        nws = int(_hip.load().pcfa_deconv4s2_fewout_workspace_bytes(B, K, N, H, W))
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32) if nws else None
        _call("pcfa_deconv4s2_fewout_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), B, K, N, H, W)
        ctx.save_for_backward(w)
        ctx.dims = (B, K, N, H, W)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("deconv4s2_fewout is the frozen-weight path: no weight / bias gradient")
        (w,) = ctx.saved_tensors
        B, K, N, H, W = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_deconv4s2_fewout_bwd", _ptr(g), _ptr(w), _ptr(gx), B, K, N, H, W)
        return gx, None, None


def deconv4s2_fewout(x, weight, bias=None):
    """conv_transpose2d(x, weight, bias, stride=2, padding=1) for a frozen 4x4 weight with at most 4 output channels:
    PWC-Net's deconv / upfeat layers (PWCNet.py:42-43) as a streaming kernel with a fixed summation order."""
    return _Deconv4s2FewOut.apply(x, weight, bias)


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor, mul):
        _dev(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, C, factor * H, factor * W), device=x.device, dtype=torch.float32)
        _call("pcfa_upsample_bilinear_fwd", _ptr(x), _ptr(out), B * C, H, W, int(factor), float(mul))
        ctx.dims = (B, C, H, W, int(factor), float(mul))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, C, H, W, factor, mul = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, C, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_upsample_bilinear_bwd", _ptr(g), _ptr(gx), B * C, H, W, factor, mul)
        return gx, None, None


def upsample_bilinear(x, factor, mul=1.0):
    """mul * nn.Upsample(scale_factor=factor, mode='bilinear')(x) (PWCNet.py:73,321); gather backward (no atomics)."""
    return _UpsampleBilinear.apply(x, int(factor), float(mul))


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


_SIDE_STREAMS = {}


def side_stream(device):
    """The second stream of `device` on which nets/raft.py runs the context encoder beside the feature encoder (one per
    device, created on first use); scratch buffers are kept per (device, main | side)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device)
    return s


def _conv_workspace(device, nbytes):
    """One scratch buffer per (device, stream), grown on demand OUTSIDE graph captures (every capture in this package
    follows eager warm-up calls of the same shapes); convolutions are stream-ordered per stream, and two streams (the
    encoders running side by side, nets/raft.py) never share a buffer."""
    dev_idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _SIDE_STREAMS.get(dev_idx)
    idx = (dev_idx, side is not None and torch.cuda.current_stream(device) == side)
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


DENSE_BLOCK_FUSED_MASKS = True   # False: one pcfa_leaky_relu_bwd launch per layer (A/B in tools, parity tests)


class _DenseBlock(torch.autograd.Function):
    """x_{i+1} = cat(leaky_relu(conv3x3_i(x_i)), x_i) for i = 0..n-1 (PWC-Net's DenseNet decoders, PWCNet.py:234-323)
    written into ONE pre-allocated buffer: every convolution reads the channel suffix it needs in place and writes
    its output in front of it, so no torch.cat copies the growing tensor (5 copies of up to 69 MB per level).
    Batch size 1 only (a channel suffix of an NCHW tensor is contiguous only then)."""

    @staticmethod
    def forward(ctx, x0, slope, *wb):
        weights, biases = wb[0::2], wb[1::2]
        _dev(x0, *weights)
        x0 = x0.contiguous()
        B, K0, H, W = x0.shape
        if B != 1:
            raise ValueError("dense_block: batch size 1 only")
        widths = [w.shape[0] for w in weights]
        total = K0 + sum(widths)
        buf = torch.empty((1, total, H, W), device=x0.device, dtype=torch.float32)
        plane = H * W
        start = total - K0
        buf[:, start:].copy_(x0)
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
        if any(ctx.needs_input_grad[2:]):
            raise RuntimeError("dense_block is the frozen-weight path: no weight / bias gradient")
        (buf,) = ctx.saved_tensors
        total, K0, H, W = ctx.dims
        plane = H * W
        gb = g.contiguous().clone()  # running gradient of the buffer: every layer adds its input gradient to a suffix
        npk = len(ctx.packs)
        for i in range(npk - 1, -1, -1):
            bwd, k, n, start = ctx.packs[i]
            if i == npk - 1 or not DENSE_BLOCK_FUSED_MASKS:
                # LeakyReLU backward of this layer's output (the top layer's gradient arrives from outside only)
                gm = torch.empty((1, n, H, W), device=g.device, dtype=torch.float32)
                _call("pcfa_leaky_relu_bwd", _ptr_off(buf, (start - n) * plane), _ptr_off(gb, (start - n) * plane),
                      _ptr(gm), ctx.slope, n * plane)
                gm_ptr = _ptr(gm)
            else:   # already multiplied by the layer above (below): its slot of the running gradient IS the masked gradient
                gm_ptr = _ptr_off(gb, (start - n) * plane)
            # the layer's input gradient is added to the running gradient in the convolution's epilogue, in place (every
            # output element reads its own addend): no separate gradient tensor, no add launch.  The first channels of
            # the suffix are the output of layer i - 1, and this is the last contribution to their gradient: its
            # LeakyReLU backward rides in the same epilogue (mask = that layer's output in the block buffer, applied
            # after the addend) -- 4 elementwise launches less per block.
            dst = _ptr_off(gb, start * plane)
            if i > 0 and DENSE_BLOCK_FUSED_MASKS:
                _conv3x3_run(g.device, gm_ptr, bwd, None, _ptr_off(buf, start * plane), dst, dst, 1, n, k, H, W,
                             slope=ctx.slope, mask_channels=ctx.packs[i - 1][2])
            else:
                _conv3x3_run(g.device, gm_ptr, bwd, None, None, dst, dst, 1, n, k, H, W)
        return (gb[:, total - K0:], None) + (None,) * (2 * len(ctx.packs))


def dense_block(x, layers, slope=0.1):
    """layers = [(weight, bias), ...] of frozen 3x3 convolutions; returns cat(y_n-1, ..., y_0, x) along channels."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _DenseBlock.apply(x, slope, *flat)


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


# --------------------------------------------------------------------------- #
# GMA attention (models/gma/gma.py:34-77,79-115)
# --------------------------------------------------------------------------- #
def gemm_f32(a, b, a_kmajor, b_kmajor, alpha=1.0, splits=1, out=None):
    """C[..., m, n] = alpha * sum_k A(m, k) B(k, n) on the fp32 matrix cores (pcfa_gemm_f32).  `a` is [.., M, K]
    (a_kmajor = 0) or [.., K, M] (1); `b` is [.., N, K] (b_kmajor = 0) or [.., K, N] (1); leading dims = batch."""
    _dev(a, b)
    a, b = a.contiguous(), b.contiguous()
    M, K = (a.shape[-1], a.shape[-2]) if a_kmajor else (a.shape[-2], a.shape[-1])
    N = b.shape[-1] if b_kmajor else b.shape[-2]
    if (b.shape[-2] if b_kmajor else b.shape[-1]) != K or a.shape[:-2] != b.shape[:-2]:
        raise ValueError("gemm_f32: operand shapes %s / %s do not match" % (tuple(a.shape), tuple(b.shape)))
    batch = 1
    for d in a.shape[:-2]:
        batch *= d
    if out is None:
        out = torch.empty(a.shape[:-2] + (M, N), device=a.device, dtype=torch.float32)
    lib = _hip.load()
    ws, nbytes = None, 0
    if splits > 1:
        nbytes = int(lib.pcfa_gemm_f32_workspace_bytes(M, N, batch, splits))
        ws = torch.empty(nbytes // 4, device=a.device, dtype=torch.float32)
    _call("pcfa_gemm_f32", _ptr(a), _ptr(b), _ptr(out), M, N, K, a.shape[-1], b.shape[-1], N, int(a_kmajor),
          int(b_kmajor), batch, M * K, N * K, M * N, float(alpha), int(splits), _ptr(ws), ctypes.c_size_t(nbytes))
    return out


GMA_GEMM = os.environ.get("PCFA_GMA_GEMM", "lib")   # "lib" | "hip": read once at import; tests assign the attribute


def _attn_mm(a, b, a_kmajor, b_kmajor, alpha=1.0, splits=1):
    """A plain GEMM of the attention block.  Default: the library (rocBLAS through torch.matmul) -- these are plain
    dense products and it runs them at 107-126 TFLOP/s; PCFA_GMA_GEMM=hip routes them through pcfa_gemm_f32 (80-105
    TFLOP/s, tools/bench_gemm.py), which the parity test exercises either way."""
    if GMA_GEMM == "hip":
        return gemm_f32(a, b, a_kmajor, b_kmajor, alpha=alpha, splits=splits)
    at = a.transpose(-1, -2) if a_kmajor else a
    bt = b if b_kmajor else b.transpose(-1, -2)
    out = torch.matmul(at, bt)
    return out if alpha == 1.0 else out.mul_(alpha)


class _AttentionSoftmax(torch.autograd.Function):
    """attn = softmax(scale * q k^T) (gma.py:52-74, content-only branch): the similarity product (plain GEMM), then the
    row softmax as ONE read and ONE write of the [N, N] matrix, in place (pcfa_softmax_rows_fwd: a 28 KB row lives in
    the registers of one workgroup; the library makes three passes), and the same in the backward: d sim = attn * (g -
    rowsum(g * attn)) in one pass, dq = scale * dsim k, dk = scale * dsim^T q."""

    @staticmethod
    def forward(ctx, q, k, scale):
        _dev(q, k)
        q, k = q.contiguous(), k.contiguous()
        sim = _attn_mm(q, k, 0, 0, alpha=scale)                       # [.., N, N]
        n = sim.shape[-1]
        _call("pcfa_softmax_rows_fwd", _ptr(sim), _ptr(sim), sim.numel() // n, n)
        ctx.scale = float(scale)
        ctx.save_for_backward(q, k, sim)
        return sim

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        q, k, attn = ctx.saved_tensors
        g = g.contiguous()
        n = attn.shape[-1]
        # never in place on `g`: autograd forbids mutating a gradient it hands in (a hook, retain_grad() on the attention
        # matrix or a second consumer would see the overwritten values).  Same traffic either way (one read of attn and
        # g, one write); the price is a 198 MB temporary at 55x128.
        ds = torch.empty_like(g)
        _call("pcfa_softmax_rows_bwd", _ptr(attn), _ptr(g), _ptr(ds), attn.numel() // n, n)
        dq = _attn_mm(ds, k, 0, 1, alpha=ctx.scale, splits=8) if ctx.needs_input_grad[0] else None    # dsim k
        dk = _attn_mm(ds, q, 1, 1, alpha=ctx.scale, splits=8) if ctx.needs_input_grad[1] else None    # dsim^T q
        return dq, dk, None


def attention_softmax(q, k, scale):
    """softmax(scale * q k^T, dim=-1) for q, k [.., N, d]."""
    return _AttentionSoftmax.apply(q, k, scale)


class AttnGradShare:
    """One attention matrix multiplied by a different value tensor in every refinement iteration (gma.py:79-115 called
    from update.py:128-130): its gradient is sum_i g_i v_i^T.  The nodes park (g_i, v_i); whichever runs last forms
    ONE product [g_1 | .. | g_n] [v_1 | .. | v_n]^T (K = n * 128) instead of n read-modify-write products over the
    198 MB matrix."""

    def __init__(self):
        self.pending = 0
        self.gs, self.vs = [], []


class _AttnTimesValue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, v, shared):
        _dev(attn, v)
        v = v.contiguous()
        ctx.save_for_backward(attn, v)
        ctx.shared = shared
        shared.pending += 1
        return _attn_mm(attn, v, 0, 1, splits=8)                       # [.., N, d]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        attn, v = ctx.saved_tensors
        sh = ctx.shared
        g = g.contiguous()
        dv = _attn_mm(attn, g, 1, 1, splits=8) if ctx.needs_input_grad[1] else None     # attn^T g
        d_attn = None
        if ctx.needs_input_grad[0]:
            if sh.pending <= 0:
                raise RuntimeError("GMA attention gradient: backward re-entered after the shared buffers were released; "
                                   "run a fresh forward (retain_graph is not supported on this path)")
            sh.gs.append(g)
            sh.vs.append(v)
            sh.pending -= 1
            if sh.pending == 0:
                gcat, vcat = torch.cat(sh.gs, dim=-1), torch.cat(sh.vs, dim=-1)
                sh.gs, sh.vs = [], []
                d_attn = _attn_mm(gcat, vcat, 0, 0)                    # [.., N, N], K = n * d
        return d_attn, dv, None


def attn_times_value(attn, v, shared):
    return _AttnTimesValue.apply(attn, v, shared)


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
    """Reduction scratch of the loss / metric kernels (32 KB), one per (device, stream), allocated once.
    While a hipGraph is being captured the capture stream reuses a buffer that was allocated OUTSIDE any capture
    (every capture in this package is preceded by eager warm-up calls on the same device), so no scratch comes from --
    and pins -- a graph's private memory pool; the kernels of one closure are stream-ordered on one stream at a time."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (idx, None if capturing else torch.cuda.current_stream().cuda_stream)
    ws = _WS.get(key)
    if ws is None and capturing:
        ws = next((w for (d, s_), w in _WS.items() if d == idx and s_ is not None), None)
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
