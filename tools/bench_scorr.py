#!/usr/bin/env python3
"""Micro-benchmark of the PWC-Net cost volume (pcfa_spatial_corr_* and the fused pcfa_cost_volume9_*) on the five
level shapes of one closure at 384x1280 (KITTI-15 padded) or 448x1024 (Sintel padded).  Run it under
`rocprofv3 --kernel-trace --stats` for per-kernel durations (tools/prof_kernels.sh); the wall figures printed here
include launch overhead.  usage: bench_scorr.py [kitti|sintel] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402

LEVELS = {2: 32, 3: 64, 4: 96, 5: 128, 6: 196}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "kitti"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    hp, wp = (384, 1280) if which == "kitti" else (448, 1024)
    dev = torch.device("cuda")
    tot_f = tot_b = bytes_f = bytes_b = 0.0
    dev_f, dev_b = [], []
    for lvl in (6, 5, 4, 3, 2):
        c, h, w = LEVELS[lvl], hp >> lvl, wp >> lvl
        a = torch.randn(1, c, h, w, device=dev)
        b = torch.randn(1, c, h, w, device=dev)
        go = torch.randn(1, 81, h, w, device=dev)
        for fused in (False, True):
            if fused:
                fwd = lambda: hip_ops._PwcCostVolume.apply(a_, b_, 0.1)  # noqa: E731
            else:
                fwd = lambda: hip_ops.spatial_correlation_sample(a_, b_, 1, 9, 1).view(1, 81, h, w)  # noqa: E731
            a_, b_ = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
            for _ in range(5):
                fwd().backward(go)
            torch.cuda.synchronize()
            with torch.no_grad():
                t0 = time.perf_counter()
                for _ in range(reps):
                    fwd()
                torch.cuda.synchronize()
                tf = (time.perf_counter() - t0) / reps * 1e6
            out = fwd()
            t0 = time.perf_counter()
            for _ in range(reps):
                out.backward(go, retain_graph=True)
            torch.cuda.synchronize()
            tb = (time.perf_counter() - t0) / reps * 1e6
            if fused and os.environ.get("PCFA_BENCH_NO_TRACER", "0") != "1":
                from torch.autograd import DeviceType
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    for _ in range(20):
                        o2 = fwd()
                        o2.backward(go)
                    torch.cuda.synchronize()
                acc = {}
                for ev in prof.events():
                    if ev.device_type == DeviceType.CUDA and "scorr9" in ev.name:
                        k = "fwd" if "fwd" in ev.name else "bwd"
                        acc.setdefault(k, []).append(ev.time_range.elapsed_us())
                kf, kb = sum(acc["fwd"]) / len(acc["fwd"]), sum(acc["bwd"]) / len(acc["bwd"])
                print("      device time per launch: fwd %.2f us, bwd (both gradients) %.2f us" % (kf, kb))
                dev_f.append(kf)
                dev_b.append(kb)
            bf = (2 * c + 81) * h * w * 4
            bb = (81 + 4 * c) * h * w * 4
            print("L%d %3dx%-4d C=%3d %-6s fwd %6.1f us wall (%5.2f TB/s)  bwd %6.1f us wall (%5.2f TB/s)"
                  % (lvl, h, w, c, "fused" if fused else "plain", tf, bf / tf / 1e6, tb, bb / tb / 1e6))
            if fused:
                tot_f += tf
                tot_b += tb
                bytes_f += bf
                bytes_b += bb
    if dev_f:
        print("sum of 5 levels, DEVICE time (HIP activity tracer): fwd %.1f us = %.2f TB/s = %.1f %% of 8 TB/s; "
              "bwd %.1f us = %.2f TB/s = %.1f %%" % (sum(dev_f), bytes_f / sum(dev_f) / 1e6, bytes_f / sum(dev_f) / 8e4,
                                                   sum(dev_b), bytes_b / sum(dev_b) / 1e6, bytes_b / sum(dev_b) / 8e4))
    print("sum of 5 levels (fused, wall incl. launch overhead): fwd %.1f us for %.1f MB, bwd %.1f us for %.1f MB"
          % (tot_f, bytes_f / 1e6, tot_b, bytes_b / 1e6))


if __name__ == "__main__":
    main()
