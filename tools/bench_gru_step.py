#!/usr/bin/env python3
"""Device time of one SepConvGRU update (models/raft/update.py:33-60: both half-steps, forward + backward = 8 gate
convolutions with their fused GRU epilogues + 1 elementwise launch) at the bench shape, with the direct kernels and with
the 1-D Winograd F(2,5) kernels (pcfa_sepconv5_algo), per kernel (HIP activity tracer).

    python tools/bench_gru_step.py [--shape 1,128,128,55,128] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import _hip, hip_ops  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="1,128,128,55,128")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    B, C, Cr, H, W = (int(v) for v in a.shape.split(","))
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    h, rest = torch.tanh(rnd(B, C, H, W)).requires_grad_(True), rnd(B, Cr, H, W).requires_grad_(True)
    halves = []
    for k in ((1, 5), (5, 1)):
        sc = (5 * (C + Cr)) ** -.5
        halves.append((rnd(2 * C, C + Cr, *k) * sc, rnd(B, 2 * C, H, W).requires_grad_(True), rnd(C, C + Cr, *k) * sc,
                       rnd(B, C, H, W).requires_grad_(True)))
    go = rnd(B, C, H, W)
    lib = _hip.load()
    outs = {}
    for algo, name in ((0, "direct"), (1, "winograd F(2,5)")):
        lib.pcfa_sepconv5_algo(algo)
        for _ in range(3):
            hip_ops.gru_step(h, rest, tuple(halves)).backward(go)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(a.reps):
                out = hip_ops.gru_step(h, rest, tuple(halves))
                out.backward(go)
            torch.cuda.synchronize()
        outs[name] = (out.detach().clone(), h.grad.clone())
        h.grad = None
        per = {}
        for ev in prof.events():
            if ev.device_type == DeviceType.CUDA:
                per.setdefault(ev.name, []).append(ev.time_range.elapsed_us())
        tot = sum(sum(v) for v in per.values()) / a.reps
        print("%s: %.1f us of device time per update (forward + backward), %d launches" %
              (name, tot, sum(len(v) for v in per.values()) // a.reps))
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            print("    %7.2f us x %4.1f  %s" % (sum(v) / len(v), len(v) / a.reps, k[:150]))
    (o0, g0), (o1, g1) = outs["direct"], outs["winograd F(2,5)"]
    print("winograd vs direct: output rel L2 %.2e, dh rel L2 %.2e" %
          (float((o1 - o0).norm() / o0.norm()), float((g1 - g0).norm() / g0.norm())))


if __name__ == "__main__":
    main()
