#!/usr/bin/env python3
"""Every conv3x3 shape of one closure of NET (default PWCNet 375x1242), timed under the algorithm named by
PCFA_CONV3X3_ALGO (f23 | f43; unset = policy).  Run once per algorithm, compare the tables:
  conv3x3_shapes_ab.py [NET HxW]  ->  lines "B K N H W count us"."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd import _hip, hip_ops  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "PWCNet"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "375x1242").split("x"))
dev = torch.device("cuda", 0)
st = bench.AttackStepper(net, h, w, dev, seed=0, boxconstraint="clipping", joint=True) if net == "PWCNet" else \
    bench.AttackStepper(net, h, w, dev, seed=0)
shapes = {}
orig = hip_ops._conv3x3_run


def spy(device, x_ptr, packed, bias_ptr, mask_ptr, addend_ptr, out_ptr, B, K, N, H, W, act=0, slope=0.):
    shapes[(B, K, N, H, W)] = shapes.get((B, K, N, H, W), 0) + 1
    return orig(device, x_ptr, packed, bias_ptr, mask_ptr, addend_ptr, out_ptr, B, K, N, H, W, act, slope)


hip_ops._conv3x3_run = spy
st.optimizer.zero_grad()
st._closure_body()
torch.cuda.synchronize()
hip_ops._conv3x3_run = orig
lib = _hip.load()
tot = 0.0
for (B, K, N, H, W), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][3] * kv[0][4] * kv[0][1] * kv[0][2]):
    x = torch.randn(B, K, H, W, device=dev)
    wt = torch.randn(N, K, 3, 3, device=dev) / (9 * K) ** .5
    fn = lambda: hip_ops.conv3x3(x, wt, None, False, 0.1)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    tot += us * cnt
    print("%d %d %d %d %d %d %.1f algo%d ks%d" % (B, K, N, H, W, cnt, us, lib.pcfa_conv3x3_algo(B, K, N, H, W),
                                                 lib.pcfa_f43_ksplit(B, K, N, H, W) if hasattr(lib, "pcfa_f43_ksplit") else -1))
print("total %.1f us per closure in conv3x3 launches (back-to-back timing)" % tot)
