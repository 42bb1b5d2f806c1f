#!/usr/bin/env python3
"""The four hoisted context convolutions of SepConvGRU (update.py:33-60 split by input, nets/raft.py precompute):
Conv2d(128 -> 256 / 128, (1,5) / (5,1)) on `inp` at 55x128 -- library convolution vs pcfa_sepconv5, forward + backward."""
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
x = torch.randn(1, 128, 55, 128, generator=g).to(dev).requires_grad_(True)
for cout in (256, 128):
    for k in ((1, 5), (5, 1)):
        w = (torch.randn(cout, 128, *k, generator=g) / 25).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        go = torch.randn(1, cout, 55, 128, generator=g).to(dev)
        pad = (0, 2) if k == (1, 5) else (2, 0)
        fns = {"library": lambda: F.conv2d(x, w, b, padding=pad),
               "sepconv5 + bias add": lambda: hip_ops.sepconv5(x, None, w) + b.view(1, -1, 1, 1)}
        for name, fn in fns.items():
            for _ in range(3):
                x.grad = None
                fn().backward(go)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for _ in range(10):
                    x.grad = None
                    fn().backward(go)
                torch.cuda.synchronize()
            evs = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
            tot = sum(e.time_range.elapsed_us() for e in evs) / 10
            print("cout %3d %s  %-20s %7.1f us per fwd+bwd in %d launches" % (cout, k, name, tot, len(evs) // 10))
