#!/usr/bin/env python3
"""The encoders' stem (Conv2d(3, 64, 7, stride 2, padding 3) on 440x1024, extractor.py:118) and stride-2 3x3 layers in the
library: forward + data gradient device time, kernels used."""
import torch
import torch.nn.functional as F
from torch.autograd import DeviceType
from torch.profiler import ProfilerActivity, profile

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
cases = [("stem 7x7 s2 B2", (2, 3, 440, 1024), (64, 3, 7, 7), 2, 3), ("stem 7x7 s2 B1", (1, 3, 440, 1024), (64, 3, 7, 7), 2, 3),
         ("3x3 s2 64->96 B2", (2, 64, 220, 512), (96, 64, 3, 3), 2, 1), ("3x3 s2 96->128 B2", (2, 96, 110, 256), (128, 96, 3, 3), 2, 1),
         ("1x1 s2 64->96 B2", (2, 64, 220, 512), (96, 64, 1, 1), 2, 0)]
for name, xs, ws, stride, pad in cases:
    x = torch.randn(*xs, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(*ws, generator=g) / 10).to(dev)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    go = torch.randn(y.shape, generator=g).to(dev)
    for _ in range(3):
        x.grad = None
        F.conv2d(x, w, None, stride=stride, padding=pad).backward(go)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            x.grad = None
            F.conv2d(x, w, None, stride=stride, padding=pad).backward(go)
        torch.cuda.synchronize()
    acc = {}
    for e in prof.events():
        if e.device_type == DeviceType.CUDA:
            acc.setdefault(e.name[:60], []).append(e.time_range.elapsed_us())
    tot = sum(sum(v) for v in acc.values()) / 10
    print("%-20s %7.1f us fwd+bwd: %s" % (name, tot, "; ".join("%s %.0f" % (k[:34], sum(v) / 10) for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])))))
