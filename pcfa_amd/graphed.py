"""hipGraph capture of the PCFA closure.

One closure evaluation is ~1900 kernel launches (MIOpen convolutions, small elementwise kernels, the pcfa_amd
kernels); launched eagerly from Python the GPU idles ~10 % of the time between them.  The closure has static
shapes and static addresses (L-BFGS updates the optimisation variables in place), so it is captured ONCE per
image pair into a hipGraph (torch.cuda.CUDAGraph) -- forward, loss and backward -- and replayed for every
closure evaluation of every L-BFGS iteration.  Same kernels, same order, same arithmetic: results equal the
eager path up to MIOpen's own run-to-run noise (tests/test_gpu_parity.py::test_graphed_closure_matches_eager).
"""
import torch


class GraphedClosure:
    """closure_fn() must (re)compute the loss from `params` and call .backward() on it, returning the loss.

    Precondition: no autograd graph that reaches `params` may still be alive (e.g. a loss tensor kept from an
    eager closure): its AccumulateGrad nodes remember the stream of their first backward, and the engine would
    then try to synchronise that stream with the capture stream -- hipStreamEndCapture crashes on it.
    """

    def __init__(self, closure_fn, params, warmup=2):
        self.params = list(params)
        dev = self.params[0].device
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up off the capture stream: MIOpen first-call work, workspaces
            for _ in range(warmup):
                for p in self.params:
                    p.grad = None
                closure_fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        for p in self.params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = closure_fn()
        self.grads = [p.grad for p in self.params]
        self.replays = 0

    def __call__(self):
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        self.replays += 1
        return self.loss


class GraphedForward:
    """Capture of a no-grad forward (the re-prediction after every L-BFGS step)."""

    def __init__(self, forward_fn, device, warmup=1):
        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                forward_fn()
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = forward_fn()

    def __call__(self):
        self.graph.replay()
        return self.out
