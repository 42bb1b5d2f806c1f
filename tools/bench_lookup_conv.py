#!/usr/bin/env python3
"""Micro-benchmark at the bench shape (55x128 features, D = 256): lookup + convc1 + ReLU fused
(pcfa_lookup_convc1_fwd / _bwd) against the un-fused sequence (pcfa_corr_lookup_fwd + library 1x1 convolution + ReLU and
their backward).  Device time per launch from the HIP activity tracer."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda")
    H, W, D = 55, 128, 256
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(1, D, H, W, generator=g).to(dev).requires_grad_(True)
    f2 = torch.randn(1, D, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(256, 324, 1, 1, generator=g) / 18).to(dev)
    b = torch.zeros(256, device=dev)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    coords = (torch.stack([xs, ys], 0).float()[None] + 3 * torch.randn(1, 2, H, W, generator=g)).to(dev)
    go = torch.randn(1, 256, H, W, generator=g).to(dev)
    blk = hip_ops.CorrBlock(f1, f2)
    for name, fn in (("fused", lambda: blk.lookup_conv_relu(coords, w, b, True)),
                     ("unfused", lambda: F.relu(F.conv2d(blk(coords), w, b)))):
        for _ in range(3):
            fn().backward(go, retain_graph=True)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(20):
                fn().backward(go, retain_graph=True)
            torch.cuda.synchronize()
        acc = {}
        for ev in prof.events():
            if ev.device_type == DeviceType.CUDA and "gemm_f32_mfma" not in ev.name and "f2ext" not in ev.name \
                    and "splitk" not in ev.name:
                acc.setdefault(ev.name[:90], []).append(ev.time_range.elapsed_us())
        tot = 0.0
        print("== %s" % name)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            if len(v) >= 20:
                print("   %-90s %3d x %7.2f us" % (k, len(v) // 20, sum(v) / len(v)))
                tot += sum(v) / 20
        print("   total per forward+backward: %.1f us" % tot)


if __name__ == "__main__":
    main()
