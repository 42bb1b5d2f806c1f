#!/usr/bin/env python3
"""conv3x3 (Winograd on fp32 MFMA) against the library convolution at the update-block / encoder shapes: mean launch
time from events around back-to-back launches, forward and data gradient."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def device_us(fn, iters=20, warm=3):
    """Mean device time per call: sum of the kernel durations the HIP activity tracer records (host launch overhead --
    two launches per call on the channel-split F(4x4,3x3) path -- does not enter)."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    acc = {}
    for ev in prof.events():
        if ev.device_type == DeviceType.CUDA:
            acc[ev.name[:60]] = acc.get(ev.name[:60], 0.0) + ev.time_range.elapsed_us()
    return sum(acc.values()) / iters, {k: v / iters for k, v in acc.items()}


def main():
    shapes = [(1, 256, 192, 55, 128), (1, 256, 126, 55, 128), (1, 128, 256, 55, 128), (1, 128, 64, 55, 128),
              (1, 192, 256, 55, 128), (1, 126, 256, 55, 128), (2, 128, 128, 55, 128), (2, 96, 96, 110, 256),
              (2, 64, 64, 220, 512), (1, 64, 64, 220, 512)]
    if "--few" in sys.argv:
        shapes = [(1, 256, 192, 55, 128), (1, 64, 64, 220, 512)]
    for B, K, N, H, W in shapes:
        x = torch.randn(B, K, H, W, device=DEV)
        w = torch.randn(N, K, 3, 3, device=DEV) / (9 * K) ** .5
        b = torch.randn(N, device=DEV)
        t_mine, parts = device_us(lambda: hip_ops.conv3x3(x, w, b, True))
        t_lib = timeit(lambda: F.relu(F.conv2d(x, w, b, padding=1))) if "--no-lib" not in sys.argv else float("nan")
        gf = 2 * 9 * K * N * H * W * B
        lib = hip_ops._hip.load()
        print("conv3x3 B%d %3d->%3d %3dx%3d  F(%d) winograd-mfma %7.1f us (%6.1f eff. TFLOP/s)   library conv+relu %7.1f us" %
              (B, K, N, H, W, lib.pcfa_conv3x3_algo(B, K, N, H, W), t_mine, gf / t_mine / 1e6, t_lib))
        if "-v" in sys.argv:
            for k, v in sorted(parts.items(), key=lambda kv: -kv[1]):
                print("        %7.1f us  %s" % (v, k))


if __name__ == "__main__":
    main()
