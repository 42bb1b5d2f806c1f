#!/usr/bin/env python3
"""The encoders' stride-2 convolutions (extractor.py:118 stem, :23-58 residual-block entries) on pcfa_conv_s2_* against
the library: device time of forward and of forward + data gradient, per kernel."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402
from tools.dev.bench_conv3x3 import device_us  # noqa: E402

DEV = "cuda"
CASES = [("stem 7x7 3->64", 2, 3, 64, 7, 440, 1024), ("stem 7x7 3->64", 1, 3, 64, 7, 440, 1024),
         ("3x3 64->96", 2, 64, 96, 3, 220, 512), ("3x3 64->96", 1, 64, 96, 3, 220, 512),
         ("3x3 96->128", 2, 96, 128, 3, 110, 256), ("3x3 96->128", 1, 96, 128, 3, 110, 256)]


def main():
    for name, B, Cin, N, k, H, W in CASES:
        x = torch.randn(B, Cin, H, W, device=DEV, requires_grad=True)
        w = torch.randn(N, Cin, k, k, device=DEV) / (Cin * k * k) ** .5
        go = torch.randn(B, N, H // 2, W // 2, device=DEV)

        def mine():
            x.grad = None
            hip_ops.conv_s2(x, w).backward(go)

        def lib():
            x.grad = None
            F.conv2d(x, w, None, stride=2, padding=k // 2).backward(go)

        gf = 2 * k * k * Cin * N * (H // 2) * (W // 2) * B * 1e-9
        for tag, fn in ((("hip", mine),) if "--hip-only" in sys.argv else (("hip", mine), ("lib", lib))):
            t, parts = device_us(fn)
            print("%-16s B%d %s fwd+bwd %7.1f us (%5.1f GFLOP per direction)" % (name, B, tag, t, gf))
            for kname, us in sorted(parts.items(), key=lambda kv: -kv[1]):
                print("      %7.1f us  %s" % (us, kname))


def pair():
    """conv1 + downsample[0] of the stride-2 blocks: fused launch against conv_s2 + library 1x1."""
    for B, Cin, N, H, W in [(2, 64, 96, 220, 512), (1, 64, 96, 220, 512), (2, 96, 128, 110, 256), (1, 96, 128, 110, 256)]:
        x = torch.randn(B, Cin, H, W, device=DEV, requires_grad=True)
        w = torch.randn(N, Cin, 3, 3, device=DEV) / (Cin * 9) ** .5
        wd = torch.randn(N, Cin, 1, 1, device=DEV) / Cin ** .5
        go = torch.randn(B, N, H // 2, W // 2, device=DEV)

        def fused():
            x.grad = None
            y, yd = hip_ops.conv_s2_ds(x, w, wd)
            (y * go).sum().backward() if False else torch.autograd.backward([y, yd], [go, go])

        def split():
            x.grad = None
            y, yd = hip_ops.conv_s2(x, w), F.conv2d(x, wd, None, stride=2)
            torch.autograd.backward([y, yd], [go, go])

        for tag, fn in (("fused", fused), ("split", split)):
            t, parts = device_us(fn)
            print("block entry %d->%d B%d %s fwd+bwd %7.1f us" % (Cin, N, B, tag, t))
            for kname, us in sorted(parts.items(), key=lambda kv: -kv[1]):
                print("      %7.1f us  %s" % (us, kname))


if __name__ == "__main__":
    if "--pair" in sys.argv:
        pair()
        sys.exit(0)
    main()
