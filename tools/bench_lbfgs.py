#!/usr/bin/env python3
"""The optimiser's vector math alone at the BASELINE size (n = 2 x 3 x 440 x 1024 floats) and a full history:
per-iteration device time of the Gram form (pcfa_lbfgs_gram_update + _direction: two sweeps over the history) against
the two-loop form (pcfa_lbfgs_pair + _direction: 2m+1 sweeps of the vector), with the HBM rate each achieves.

    python tools/bench_lbfgs.py [history=100] [iters=20]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import _hip  # noqa: E402
from pcfa_amd.hip_ops import _call, _ptr  # noqa: E402


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    lib = _hip.load()
    n = 2 * 3 * 440 * 1024
    ld, rows = n, cap + 1
    g = torch.Generator().manual_seed(0)
    S = torch.randn(rows, ld, generator=g).to(dev) * 1e-3
    Y = (S + 0.3e-3 * torch.randn(rows, ld, generator=g).to(dev)).contiguous()
    grad = torch.randn(ld, generator=g).to(dev)
    g_prev = grad - Y[0]
    d = (S[0] * 1.0).contiguous()
    out2 = torch.empty(2, device=dev)
    state = torch.zeros(int(lib.pcfa_lbfgs_gram_state_bytes(cap)), dtype=torch.uint8, device=dev)
    ws = torch.empty(max(int(lib.pcfa_lbfgs_gram_workspace_bytes(cap, ld)) // 4, int(lib.pcfa_lbfgs_workspace_floats())),
                     device=dev)
    _call("pcfa_lbfgs_gram_reset", _ptr(state), cap)

    def gram_iter():
        _call("pcfa_lbfgs_gram_update", _ptr(grad), _ptr(g_prev), _ptr(d), 1.0, _ptr(S), _ptr(Y), _ptr(state), _ptr(ws),
              cap, ld)
        _call("pcfa_lbfgs_gram_direction", _ptr(grad), _ptr(S), _ptr(Y), _ptr(state), _ptr(d), _ptr(out2), _ptr(ws),
              cap, ld)

    # fill the history: every update sees g = g_prev + (s + noise) with s = d, so y.s > 0 and the pair is kept
    for k in range(cap + 3):
        grad.copy_(g_prev + Y[(k * 7) % rows] * 1.0 + 1e-3 * torch.randn(ld, device=dev))
        gram_iter()
        d.copy_(S[(k * 5) % rows])        # keep d benign (the direction itself is not the point here)
    hdr = state[:16].view(torch.int32).tolist()
    print("history after fill: first %d count %d accepted %d" % (hdr[0], hdr[1], hdr[2]))

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def gram_update_only():
        _call("pcfa_lbfgs_gram_update", _ptr(grad), _ptr(g_prev), _ptr(d), 1.0, _ptr(S), _ptr(Y), _ptr(state), _ptr(ws),
              cap, ld)
        d.copy_(S[3])

    def gram_direction_only():
        _call("pcfa_lbfgs_gram_direction", _ptr(grad), _ptr(S), _ptr(Y), _ptr(state), _ptr(d), _ptr(out2), _ptr(ws),
              cap, ld)

    # keep the pair acceptable while timing the update alone: restore grad/g_prev relation each call
    base = grad.clone()

    def upd():
        g_prev.copy_(base - Y[5])
        d.copy_(S[5])
        grad.copy_(base)
        _call("pcfa_lbfgs_gram_update", _ptr(grad), _ptr(g_prev), _ptr(d), 1.0, _ptr(S), _ptr(Y), _ptr(state), _ptr(ws),
              cap, ld)

    def copies():
        g_prev.copy_(base - Y[5])
        d.copy_(S[5])
        grad.copy_(base)

    t_copies = timed(copies)
    t_upd = timed(upd) - t_copies
    t_dir = timed(gram_direction_only)
    m = state[:16].view(torch.int32).tolist()[1]
    hist_bytes = 2 * m * ld * 4
    print("gram   m=%3d: update %.3f ms (%.2f TB/s over the history)   direction %.3f ms (%.2f TB/s)   = %.3f ms / iter"
          % (m, t_upd, hist_bytes / t_upd / 1e9, t_dir, hist_bytes / t_dir / 1e9, t_upd + t_dir))

    # two-loop form on the same ring
    ro = torch.zeros(rows, device=dev)
    for r in range(rows):
        ro[r] = 1.0 / (Y[r] @ S[r])
    H = torch.ones(1, device=dev)
    al = torch.empty(rows, device=dev)
    scal = torch.empty(4, device=dev)

    def two_loop():
        _call("pcfa_lbfgs_pair", _ptr(grad), _ptr(g_prev), _ptr(d), 1.0, _ptr(Y[cap]), _ptr(S[cap]), _ptr(scal),
              _ptr(ws), 1, n)
        _call("pcfa_lbfgs_direction", _ptr(grad), _ptr(S), _ptr(Y), _ptr(ro), _ptr(H), _ptr(al), _ptr(d), _ptr(ws), 0,
              cap, rows, ld, n)

    t2 = timed(two_loop)
    print("two-loop m=%3d: %.3f ms / iter (%d launches, %.1f GB)" % (cap, t2, 2 * cap + 3,
                                                                      (2 * cap + 1) * 4 * n * 4 / 1e9))


if __name__ == "__main__":
    main()
