#!/usr/bin/env python3
"""pwc_warp deterministic backward: run once per scatter form (PCFA_WARP_SCATTER=global: one atomic per tap; default: the LDS
window) or per variant library -- originally with and without the in-register combination of neighbouring lanes' taps
(-DPCFA_WARP_COMBINE=0 variant: tools/dev/build_variant.sh warp0 warp_ops.hip -DPCFA_WARP_COMBINE=0, selected with
PCFA_HIP_LIB): device time per call at the four KITTI-size levels (smooth, PWC-like and noisy flow) and an exact checksum
of both gradients -- the two builds must print the same checksums (integer addends: the combination moves no bit).
usage: warp_combine_ab.py   (run once per library)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402
from tools.dev.bench_conv3x3 import device_us  # noqa: E402


def chk(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())


dev = "cuda"
print("library:", os.environ.get("PCFA_HIP_LIB", "(default)"))
for shape, scale in (((1, 32, 96, 320), 6.0), ((1, 64, 48, 160), 3.0), ((1, 96, 24, 80), 1.5), ((1, 128, 12, 40), 0.8),
                     ((2, 5, 7, 9), 2.0)):
    g = torch.Generator().manual_seed(0)
    B, C, H, W = shape
    x = torch.randn(*shape, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(*shape, generator=g).to(dev)
    for kind in ("smooth", "pwc-like", "noisy"):
        if kind == "smooth":
            f = scale * torch.randn(B, 2, 1, 1, generator=g).expand(B, 2, H, W)
        elif kind == "pwc-like":   # a smooth field + 0.2 px of texture, as an up-sampled coarse flow is
            base = torch.nn.functional.interpolate(scale * torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g),
                                                   size=(H, W), mode="bilinear", align_corners=False)
            f = base + 0.2 * torch.randn(B, 2, H, W, generator=g)
        else:
            f = scale * torch.randn(B, 2, H, W, generator=g)
        flo = f.contiguous().to(dev).requires_grad_(True)
        out = hip_ops.pwc_warp(x, flo, deterministic=True, flow_scale=1.25)
        gx, gf = torch.autograd.grad(out, (x, flo), go, retain_graph=True)
        t, parts = device_us(lambda: torch.autograd.grad(out, (x, flo), go, retain_graph=True))
        scat = [v for k, v in parts.items() if "bwd_det" in k]
        print("%-18s %-9s %7.1f us (scatter %6.1f)  checksum grad_x %d  grad_flo %d" % (shape, kind, t, scat[0] if scat else -1,
                                                                                      chk(gx), chk(gf)))
