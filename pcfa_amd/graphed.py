"""hipGraph capture of the PCFA closure.

One closure evaluation is ~1900 kernel launches (MIOpen convolutions, small elementwise kernels, the pcfa_amd
kernels); launched eagerly from Python the GPU idles ~10 % of the time between them.  The closure has static
shapes and static addresses (L-BFGS updates the optimisation variables in place), so it is captured ONCE per
image pair into a hipGraph (torch.cuda.CUDAGraph) -- forward, loss and backward -- and replayed for every
closure evaluation of every L-BFGS iteration.  Same kernels, same order, same arithmetic: results equal the
eager path up to MIOpen's own run-to-run noise (tests/test_gpu_parity.py::test_graphed_closure_matches_eager).
"""
import contextlib
import gc

import torch

from .ops import core as _core


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a capture is open: if the collector frees an older pair's CUDAGraph in the
    middle of a capture, its destructor calls into HIP ("operation not permitted when stream is capturing") and the
    process aborts.  Garbage is collected before the capture instead."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _release(stream):
    """A dying graph object hands its capture stream back (ops.core.release_stream); nothing at interpreter shutdown."""
    try:
        if stream is not None:
            _core.release_stream(stream)
    except Exception:  # noqa: BLE001
        pass


class GraphedClosure:
    """closure_fn() must (re)compute the loss from `params` and call .backward() on it, returning the loss.

    Precondition: no autograd graph that reaches `params` may still be alive (e.g. a loss tensor kept from an
    eager closure): its AccumulateGrad nodes remember the stream of their first backward, and the engine would
    then try to synchronise that stream with the capture stream -- hipStreamEndCapture crashes on it.
    """

    def __init__(self, closure_fn, params, warmup=2, grad_sink=None):
        """grad_sink: one tensor per parameter (views of the optimiser's flat gradient buffer,
        `pcfa_amd.lbfgs.LBFGS.flat_grad_views`): the copies gradient -> flat buffer are captured behind the backward
        pass and `p.grad` is pointed at the views, so the optimiser finds its flat gradient in place after a replay."""
        self.params = list(params)
        dev = self.params[0].device
        cur = torch.cuda.current_stream(dev)
        side = _core.new_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up off the capture stream: MIOpen first-call work, workspaces
            for _ in range(warmup):
                for p in self.params:
                    p.grad = None
                closure_fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        _core.release_stream(side)
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        # a capture stream of this object's own: torch.cuda.graph's default is ONE stream per process, and the library
        # GEMMs' workspace is kept per (handle, stream) -- two closures captured on it (two pairs in flight,
        # attack_PCFA.PairsInFlight) would replay against the same scratch side by side.  Kept until the graph dies.
        self._capture_stream = _core.new_stream(dev)
        with _no_gc(), torch.cuda.graph(self.graph, stream=self._capture_stream):
            self.loss = closure_fn()
            if grad_sink is not None:
                with torch.no_grad():
                    for p, v in zip(self.params, grad_sink):
                        v.copy_(p.grad)
        self.grads = [p.grad for p in self.params] if grad_sink is None else list(grad_sink)
        self.replays = 0

    def __call__(self):
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        self.replays += 1
        return self.loss

    def __del__(self):
        _release(getattr(self, "_capture_stream", None))


class GraphedForward:
    """Capture of a no-grad forward (the re-prediction after every L-BFGS step)."""

    def __init__(self, forward_fn, device, warmup=1):
        cur = torch.cuda.current_stream(device)
        side = _core.new_stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                forward_fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        _core.release_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self._capture_stream = _core.new_stream(device)   # (as GraphedClosure)
        with _no_gc(), torch.no_grad(), torch.cuda.graph(self.graph, stream=self._capture_stream):
            self.out = forward_fn()

    def __call__(self):
        self.graph.replay()
        return self.out

    def __del__(self):
        _release(getattr(self, "_capture_stream", None))


class SplitGraphedClosure:
    """The closure captured as TWO graphs sharing one memory pool: F = forward + loss (autograd recording on), B = backward.

    Why: every attack step ends with a re-prediction at the updated variables (attack_PCFA.py:194-196) and the next
    step starts with a closure evaluation at exactly those variables (torch.optim.LBFGS evaluates the closure first).
    The reference computes that forward twice.  With the closure split, `forward()` serves the re-prediction and stays
    valid as the forward half of the next closure call, which then replays only B: the same kernels on the same data in
    the same order as F + B back to back, one forward (7 ms of a 160 ms step) less.  Results are unchanged.

    forward_fn() -> (loss, aux): computes the loss from `params` WITHOUT calling backward; `aux` (any structure of
    tensors: deltas, flow) is exposed as static outputs.  The caller must call `invalidate()` -- or simply `__call__`,
    which consumes the forward -- whenever `params` change after `forward()`."""

    def __init__(self, forward_fn, params, warmup=2):
        self.params = list(params)
        dev = self.params[0].device
        cur = torch.cuda.current_stream(dev)
        side = _core.new_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for p in self.params:
                    p.grad = None
                loss, _ = forward_fn()
                loss.backward()
                del loss
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        _core.release_stream(side)
        for p in self.params:
            p.grad = None
        pool = torch.cuda.graph_pool_handle()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # ONE capture stream for both graphs: autograd runs every backward node on the stream its forward ran on, so a
        # second capture on a stream of its own would have to synchronise with the first one's -- invalid in a capture
        cap = self._capture_stream = _core.new_stream(dev)
        with _no_gc():
            with torch.cuda.graph(self.fwd_graph, pool=pool, stream=cap):
                self.loss, self.aux = forward_fn()
            with torch.cuda.graph(self.bwd_graph, pool=pool, stream=cap):
                self.loss.backward()
        self.grads = [p.grad for p in self.params]
        self.loss_value = self.loss.detach()
        self._versions = []
        self.fresh = False          # True: F was replayed at the current values of `params` and B has not consumed it
        self.replays = 0
        self.forwards_shared = 0

    def forward(self):
        self.fwd_graph.replay()
        self.fresh = True
        self._versions = [p._version for p in self.params]
        return self.loss_value, self.aux

    def __del__(self):
        _release(getattr(self, "_capture_stream", None))

    def invalidate(self):
        self.fresh = False

    def __call__(self):
        # a forward is only reusable if nobody wrote to the variables since it ran (in-place writes bump _version);
        # other inputs of the loss (target, images) are the caller's to announce through invalidate()
        if self.fresh and any(p._version != v for p, v in zip(self.params, self._versions)):
            self.fresh = False
        if self.fresh:
            self.forwards_shared += 1
        else:
            self.fwd_graph.replay()
        self.bwd_graph.replay()
        self.fresh = False
        for p, g in zip(self.params, self.grads):
            p.grad = g
        self.replays += 1
        return self.loss_value
