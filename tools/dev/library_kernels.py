#!/usr/bin/env python3
"""Full names of the kernels of one eager closure that are NOT this package's (rocBLAS / Tensile, MIOpen, aten):
library_kernels.py [NET] [HxW]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

from torch.autograd import DeviceType  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128x160").split("x"))
st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
for _ in range(2):
    st.optimizer.zero_grad()
    st._closure_body()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    st.optimizer.zero_grad()
    st._closure_body()
    torch.cuda.synchronize()
acc = {}
for ev in prof.events():
    if ev.device_type == DeviceType.CUDA and "anonymous namespace" not in ev.name:
        a = acc.setdefault(ev.name, [0, 0.0])
        a[0] += 1
        a[1] += ev.time_range.elapsed_us()
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%3d %8.1f us  %s" % (v[0], v[1], k))
