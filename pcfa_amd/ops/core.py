"""Plumbing shared by every operator module: launches through the C-ABI on torch's current stream (no CPU fallback),
the optional per-launch timers and the work recorder that bench.py's roofline rows read."""
import contextlib
import ctypes
import threading

import torch

from .. import _hip

# ---- lanes: several image pairs in flight on one GPU ---------------------------------------------------------------------
# Every scratch buffer that is shared between launches (the convolutions' split-K partials, the loss kernels' reduction
# scratch) assumes ONE stream-ordered sequence of launches.  Two attacks running side by side (attack_PCFA.PairsInFlight:
# one host thread + one stream + one set of hipGraphs per pair) would race on them, so those buffers are keyed by the
# caller's LANE as well: a small integer, thread-local, 0 unless `with lane(k):` says otherwise.  Graph sets are cached per
# lane for the same reason (attack_PCFA.PairAttack.graph_key).
_lane_state = threading.local()
_STREAM_LANE = {}   # HIP stream handle -> lane.  autograd runs every backward node on ITS OWN thread (thread-locals of the
                    # caller are not there) but on the stream of the node's forward: the stream is what carries the lane


def current_lane():
    if _STREAM_LANE and torch.cuda.is_available():
        k = _STREAM_LANE.get(torch.cuda.current_stream().cuda_stream)
        if k is not None:
            return k
    return getattr(_lane_state, "k", 0)


@contextlib.contextmanager
def lane(k):
    prev = getattr(_lane_state, "k", 0)
    _lane_state.k = int(k)
    try:
        yield
    finally:
        _lane_state.k = prev


def bind_stream(stream, k=None):
    """Register `stream` (a torch.cuda.Stream) with lane k (default: the caller's lane); returns the stream.  Every stream a
    lane's work may run on -- its own, and the warm-up / capture streams of its hipGraphs (pcfa_amd/graphed.py) -- is bound,
    so that launches issued from autograd's backward thread find the lane through the stream they run on."""
    k = getattr(_lane_state, "k", 0) if k is None else int(k)
    if k != 0:
        _STREAM_LANE[stream.cuda_stream] = k
    else:
        _STREAM_LANE.pop(stream.cuda_stream, None)   # lane 0 is the default; a recycled handle must not keep an old binding
    return stream


_hiprt = None
_FREE_STREAMS = {}   # device index -> HIP stream handles of own_stream() objects that have died (reused before new ones are made)


def _hip_runtime():
    global _hiprt
    if _hiprt is None:
        rt = ctypes.CDLL("libamdhip64.so")
        rt.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        rt.hipStreamCreateWithFlags.restype = ctypes.c_int
        _hiprt = rt
    return _hiprt


def release_stream(stream):
    """Hand an own_stream() back: its handle loses its lane binding and waits on the free list for the next owner.  Called
    by the OWNER when it is done with the stream (graphed.* after the warm-up / when the graph dies, PairsInFlight when it
    dies) -- explicitly, because torch's stream objects cannot carry a weak reference (a weakref.finalize on an
    ExternalStream crashed the interpreter's final collection in _PyWeakref_ClearRef, r05)."""
    try:
        handle = stream.cuda_stream
        idx = stream.device.index
    except Exception:  # noqa: BLE001 -- interpreter shutdown
        return
    if handle in _OWN_HANDLES.get(idx, ()) and handle not in _FREE_STREAMS.setdefault(idx, []):
        _STREAM_LANE.pop(handle, None)
        _FREE_STREAMS[idx].append(handle)


_OWN_HANDLES = {}    # device index -> every handle own_stream() ever created (they are never destroyed)


def own_stream(device, k=None):
    """A stream NO OTHER live owner shares, bound to lane k (default: the caller's lane).

    `torch.cuda.Stream()` does not create a stream: it hands out one of 32 pooled HIP streams per device, round-robin, so
    the 33rd object IS the first one again while that may still be in use.  Lanes tell their work apart by the stream
    (`_STREAM_LANE`, the scratch keys of ops.conv / ops.attack_math), so an aliased handle puts two lanes on one set of
    scratch buffers: with Config.overlap_encoders (a persistent side stream per lane) every flight after the pool's first
    wrap-around came out wrong (tools/dev/flight_repeat.py, r05), and seven pairs in flight would alias even without it.
    Here the stream is created through the HIP runtime (non-blocking, as the pool's are) and wrapped as an ExternalStream;
    `release_stream` puts its handle on a free list for the next owner instead of destroying it (a hipStreamDestroy could
    land inside another thread's graph capture)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _run_pool(idx)      # (the lanes' run streams take their hardware queues first: lane_run_streams)
    free = _FREE_STREAMS.setdefault(idx, [])
    if free:
        handle = free.pop()
    else:
        h = ctypes.c_void_p()
        with torch.cuda.device(idx):
            err = _hip_runtime().hipStreamCreateWithFlags(ctypes.byref(h), 1)   # hipStreamNonBlocking
        if err != 0 or not h.value:
            raise RuntimeError("hipStreamCreateWithFlags failed: %d" % err)
        handle = h.value
        _OWN_HANDLES.setdefault(idx, set()).add(handle)
    return bind_stream(torch.cuda.ExternalStream(handle, device=torch.device("cuda", idx)), k)


_LANE_RUN_STREAMS = {}   # device index -> torch's pooled streams in pool order; lane k's attack steps run on the k-th


def _run_pool(idx):
    """torch's pooled streams of device idx in pool order, the first eight touched (a null launch each) the moment the
    package first needs ANY stream of its own -- before own_stream creates one -- so that they get their hardware queues
    while nothing else competes for them."""
    pool = _LANE_RUN_STREAMS.get(idx)
    if pool is None:
        pool, seen = [], set()
        for _ in range(64):                       # two trips round torch's pool of 32: every member, in pool order
            s = torch.cuda.Stream(torch.device("cuda", idx))
            if s.cuda_stream not in seen:
                seen.add(s.cuda_stream)
                pool.append(s)
        lib = _hip.load()
        for s in pool[:8]:
            _hip.check(lib.pcfa_null_launch(ctypes.c_void_p(s.cuda_stream)), "pcfa_null_launch")
        _LANE_RUN_STREAMS[idx] = pool
    return pool


def lane_run_streams(device, n):
    """The n streams on which n pairs in flight run their steps: persistent, one per (device, lane).

    These are the only streams of the package whose kernels are meant to overlap with each other, and kernels of two
    streams overlap only if the streams sit on different hardware queues (four per device by default).  The runtime picks
    the queue when a stream is first used and keeps it.  Measured (tools/dev/flight_scaling.py, r05): run streams taken
    from own_stream's free list put two lanes on one queue (the second pair gained 4 % instead of 30 %); streams created
    fresh and touched back to back still left lanes 0 and 3 on one queue (9.68 pair-steps/s at four pairs); torch's 32
    pooled streams, created together when the process first asks for one, spread over the queues in pool order (10.70).
    So lane k runs on the k-th DISTINCT pooled stream -- drawn once, kept for the life of the process, and never aliased by
    this package, whose other streams (graph warm-up / capture, the encoders' side streams) all come from own_stream."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _run_pool(idx)
    if n > len(pool):
        raise ValueError("%d pairs in flight: at most %d lanes per device" % (n, len(pool)))
    return [bind_stream(pool[k], k) for k in range(n)]


def new_stream(device):
    """A new stream of `device` that is nobody else's (own_stream), bound to the caller's lane."""
    return own_stream(device)


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
    "pcfa_conv3x3_fewout_bwd": lambda a: _note_work("conv3x3_fewout_bwd", 4.0 * a[4] * (a[5] + a[6]) * a[7] * a[8], 0.0),
    "pcfa_relu_bwd": lambda a: _note_work("relu_bwd", 12.0 * a[3], 0.0),
    "pcfa_relu_bwd2": lambda a: _note_work("relu_bwd2", 20.0 * a[5], 0.0),
    "pcfa_add_relu_fwd": lambda a: _note_work("add_relu_fwd", 12.0 * a[3], 0.0),
    "pcfa_conv_fewin_packed_fwd": lambda a: _note_work("conv_fewin_fwd", *(2 * [2.0 * a[5] * a[9] * a[9] * a[6] * a[4] * a[7] * a[8]])),
}


_spy = None


def set_call_spy(spy):
    """spy(name, args, invoke) sees EVERY entry-point call of every operator module while set (None: off) and decides
    itself whether to run `invoke(name, *args)` -- the hook for tools (tools/dev/conv_shapes.py).  The operator modules bind
    `_call` by name at import, so patching an attribute of this module or of the `hip_ops` table intercepts nothing."""
    global _spy
    _spy = spy


def _call(name, *args):
    """Invoke C-ABI entry point `name` on torch's current stream and raise on a non-zero status."""
    if _spy is not None:
        return _spy(name, args, _invoke)
    return _invoke(name, *args)


def _invoke(name, *args):
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


def work_recorder():
    """The dict set_work_recorder() installed, or None."""
    return _work
