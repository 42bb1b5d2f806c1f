#!/usr/bin/env python3
"""pwc_warp backward: fp32-atomics scatter vs the fixed-point (deterministic) scatter at the four KITTI-size levels,
smooth and noisy flow; device time per call (HIP activity tracer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402
from tools.dev.bench_conv3x3 import device_us  # noqa: E402

dev = "cuda"
for shape, scale in (((1, 32, 96, 320), 6.0), ((1, 64, 48, 160), 3.0), ((1, 96, 24, 80), 1.5), ((1, 128, 12, 40), 0.8)):
    g = torch.Generator().manual_seed(0)
    B, C, H, W = shape
    x = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(*shape, generator=g).to(dev)
    for kind in ("smooth", "noisy"):
        f = scale * torch.randn(B, 2, 1, 1, generator=g).expand(B, 2, H, W) if kind == "smooth" else scale * torch.randn(B, 2, H, W, generator=g)
        flo = f.contiguous().to(dev).requires_grad_(True)
        for det in (False, True):
            out = hip_ops.pwc_warp(x, flo, deterministic=det)
            t, parts = device_us(lambda: torch.autograd.grad(out, (x, flo), go, retain_graph=True))
            print("%-18s %-6s %s  %7.1f us   %s" % (shape, kind, "fixed-point" if det else "fp32 atomics", t,
                                                   "  ".join("%s %.1f" % (k.split("(")[0][-28:], v) for k, v in parts.items())))
