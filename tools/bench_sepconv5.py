#!/usr/bin/env python3
"""Device time of the SepConvGRU gate convolutions at the bench shape (55x128): z|r (Cout 256), q (Cout 128), and the
two data gradients, horizontal and vertical (HIP activity tracer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    dev = torch.device("cuda")
    H, W = 55, 128
    g = torch.Generator().manual_seed(0)
    for vertical in (False, True):
        for cout in (256, 128):
            a = torch.randn(1, 128, H, W, generator=g).to(dev).requires_grad_(True)
            b = torch.randn(1, 128, H, W, generator=g).to(dev).requires_grad_(True)
            w = (torch.randn((cout, 256) + ((5, 1) if vertical else (1, 5)), generator=g) / 36).to(dev)
            go = torch.randn(1, cout, H, W, generator=g).to(dev)
            for _ in range(3):
                hip_ops.sepconv5(a, b, w).backward(go)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(20):
                    hip_ops.sepconv5(a, b, w).backward(go)
                torch.cuda.synchronize()
            d = [ev.time_range.elapsed_us() for ev in prof.events()
                 if ev.device_type == DeviceType.CUDA and "sepconv5_kernel" in ev.name]
            fwd, bwd = d[0::2], d[1::2]
            fl = 2.0 * 5 * 256 * cout * H * W
            print("%s Cout %3d: forward %6.2f us (%5.1f TFLOP/s)   data gradient %6.2f us (%5.1f TFLOP/s)"
                  % ("5x1" if vertical else "1x5", cout, sum(fwd) / len(fwd), fl / (sum(fwd) / len(fwd)) / 1e6,
                     sum(bwd) / len(bwd), fl / (sum(bwd) / len(bwd)) / 1e6))


if __name__ == "__main__":
    main()
