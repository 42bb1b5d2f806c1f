#!/usr/bin/env python3
"""1x1 convolutions of the RAFT encoders / mask head at 55x128: library convolution vs pcfa_gemm_f32 (forward + data
gradient), device time from the HIP activity tracer."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402
from torch.autograd import DeviceType  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)


def conv_gemm(x, w, b):
    B, K, H, W = x.shape
    y = hip_ops.gemm_f32(w.view(1, w.shape[0], K).expand(B, -1, -1), x.view(B, K, H * W), 0, 1)
    return y.view(B, -1, H, W) + b.view(1, -1, 1, 1)


def conv_gemm_bwd(gy, w):
    B, N, H, W = gy.shape
    return hip_ops.gemm_f32(w.view(1, N, -1).expand(B, -1, -1), gy.view(B, N, H * W), 1, 1).view(B, -1, H, W)


for B, K, N in ((2, 128, 256), (1, 128, 256), (1, 256, 576)):
    x = torch.randn(B, K, 55, 128, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, 1, 1, generator=g) / K ** .5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    go = torch.randn(B, N, 55, 128, generator=g).to(dev)

    def lib():
        x.grad = None
        F.conv2d(x, w, b).backward(go)

    def mine():
        conv_gemm(x.detach(), w, b)
        conv_gemm_bwd(go, w)

    ref = F.conv2d(x, w, b)
    print("max diff fwd", float((conv_gemm(x.detach(), w, b) - ref).abs().max()))
    for name, fn in (("library", lib), ("pcfa_gemm_f32", mine)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
        print("B%d %d->%d %-14s %7.1f us per fwd+bwd in %d launches" % (B, K, N, name, sum(e.time_range.elapsed_us() for e in evs) / 10, len(evs) // 10))
