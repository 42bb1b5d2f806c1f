#!/usr/bin/env python3
"""Micro-benchmark of FlowNet2's three HIP operators at the BASELINE image size (436x1024 -> 448x1024):
per-kernel duration from HIP events attached to the dispatch packet (pcfa_timing_arm, the timestamps rocprofv3
reads), algorithmic bytes / FLOP per launch and the fraction of the bound that applies.

    python tools/bench_flownet_ops.py [--iters 20]

Algorithmic work per launch (B = 1, fp32):
  correlation fwd   in 2 x 256x56x128 (14.7 MB) + out 441x56x128 (12.6 MB) = 27.3 MB; 2*441*256*7168 = 1.62 GFLOP
                    (plain fp32 FMA work, not a dense contraction: 21 of 64 columns of a banded GEMM tile would be
                    useful), so the bound is the vector fp32 rate: 256 CUs x 4 SIMD x 16 lanes x 2 FLOP x 2.4 GHz = 78.6 TFLOP/s
  correlation bwd   per gradient: g 12.6 MB + other map 7.3 MB + out 7.3 MB = 27.3 MB; 1.62 GFLOP
  resample2d fwd    image 3x448x1024 (5.5 MB) + flow (3.7 MB) + out (5.5 MB) = 14.7 MB  -> HBM
  resample2d bwd    grad_out 5.5 + flow 3.7 + image 5.5 + grad_image 5.5 (zero fill) + 5.5 (atomics) + grad_flow 3.7 = 29.4 MB
  channelnorm fwd   3 planes in + 1 out = 7.3 MB;  bwd: in 5.5 + out 1.8 + grad_out 1.8 + grad_in 5.5 = 14.7 MB
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402

DEV = "cuda"
HBM_GBS = 8000.0
VALU_TFLOPS = 78.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    H, W = 448, 1024
    f1 = torch.randn(1, 256, H // 8, W // 8, generator=g).to(DEV).requires_grad_(True)
    f2 = torch.randn(1, 256, H // 8, W // 8, generator=g).to(DEV).requires_grad_(True)
    img = torch.randn(1, 3, H, W, generator=g).to(DEV).requires_grad_(True)
    flow = (8 * torch.randn(1, 2, H, W, generator=g)).to(DEV).requires_grad_(True)
    junk = torch.empty(512 * 1024 * 1024 // 4, device=DEV)  # > MALL: evicts L2 + Infinity Cache between launches

    def step(cold):
        for t in (f1, f2, img, flow):
            t.grad = None
        if cold:
            junk.add_(1.0)
        c = hip_ops.flownet_correlation(f1, f2, 20, 1, 20, 1, 2)
        if cold:
            junk.add_(1.0)
        r = hip_ops.resample2d(img, flow)
        if cold:
            junk.add_(1.0)
        n = hip_ops.channelnorm(r)
        gc, gn = torch.ones_like(c), torch.ones_like(n)
        if cold:
            junk.add_(1.0)
        (gr,) = torch.autograd.grad(n, r, gn, retain_graph=True)
        if cold:
            junk.add_(1.0)
        torch.autograd.grad(r, (img, flow), gr)
        if cold:
            junk.add_(1.0)
        torch.autograd.grad(c, (f1, f2), gc)

    px = H * W
    cpx = (H // 8) * (W // 8)
    work = {  # name -> (bytes, flop)
        "flownet_corr_fwd": (4 * (2 * 256 * cpx + 441 * cpx), 2 * 441 * 256 * cpx),
        "flownet_corr_bwd_in1": (4 * (441 * cpx + 2 * 256 * cpx), 2 * 441 * 256 * cpx),
        "flownet_corr_bwd_in2": (4 * (441 * cpx + 2 * 256 * cpx), 2 * 441 * 256 * cpx),
        "resample2d_fwd": (4 * (3 + 2 + 3) * px, 0),
        "resample2d_bwd": (4 * (3 + 2 + 3 + 3 + 3 + 2) * px, 0),
        "channelnorm_fwd": (4 * 4 * px, 0),
        "channelnorm_bwd": (4 * 8 * px, 0),
    }
    for label, cold in (("warm (back-to-back)", False), ("cold (L2 + MALL flushed before every launch)", True)):
        for _ in range(3):
            step(cold)
        timer = hip_ops.DispatchTimer()
        hip_ops.set_dispatch_timer(timer)
        try:
            for _ in range(args.iters):
                step(cold)
        finally:
            hip_ops.set_dispatch_timer(None)
        res = timer.summary()
        print("# %s, %d launches each" % (label, args.iters))
        print("%-24s %10s %12s %12s %8s" % ("kernel", "us", "GB/s (alg)", "TFLOP/s", "frac"))
        for name, (nbytes, flop) in work.items():
            if name not in res:
                continue
            us = res[name][0]
            gbs = nbytes / us / 1e3
            tf = flop / us / 1e6
            frac = tf / VALU_TFLOPS if flop else gbs / HBM_GBS
            print("%-24s %10.1f %12.0f %12.2f %7.1f%%  (%s)" % (name, us, gbs, tf, 100 * frac,
                                                              "vector fp32" if flop else "hbm"))


if __name__ == "__main__":
    main()
