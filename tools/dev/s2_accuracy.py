#!/usr/bin/env python3
"""Forward rounding error of conv_s2 and of the library convolution against an fp64 reference."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
for B, Cin, N, k, H, W in [(2, 3, 64, 7, 440, 1024), (1, 3, 64, 7, 440, 1024), (2, 64, 96, 3, 220, 512), (2, 96, 128, 3, 110, 256)]:
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(N, Cin, k, k, generator=g) / (Cin * k * k) ** .5).to(dev)
    ref = F.conv2d(x.double(), w.double(), None, stride=2, padding=k // 2)
    mine = hip_ops.conv_s2(x, w)
    lib = F.conv2d(x, w, None, stride=2, padding=k // 2)
    em, el = (mine.double() - ref).abs(), (lib.double() - ref).abs()
    print("%dx%d %d->%d k%d B%d: conv_s2 max %.2e rms %.2e | library max %.2e rms %.2e | worst conv_s2 at %s"
          % (H, W, Cin, N, k, B, em.max().item(), em.pow(2).mean().sqrt().item(), el.max().item(),
             el.pow(2).mean().sqrt().item(), tuple(int(v) for v in (em == em.max()).nonzero()[0])))
    # where are the large errors? rows / columns profile
    big = (em > 1e-5)
    print("    outputs with error > 1e-5: %d of %d; by column (first 8): %s ; by row: %s" % (
        int(big.sum()), big.numel(), big.sum(dim=(0, 1, 2))[:8].tolist(), big.sum(dim=(0, 1, 3))[:6].tolist()))
