#!/usr/bin/env python3
"""Where in the image does d(loss)/d(variables) differ between the library stride-2 layers and conv_s2 (sigma = 0)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd import hip_ops  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=0)
res = {}
for name, on in (("lib", False), ("s2", True)):
    raft.CONV_S2, hip_ops.CONV_S2_BWD, raft.FUSED_DOWNSAMPLE = on, False, False
    st.optimizer.zero_grad()
    st._closure_body()
    res[name] = [p.grad.detach().clone() for p in st.params]
for i, (a, b) in enumerate(zip(res["lib"], res["s2"])):
    d = (a - b).abs()
    print("param %d shape %s: |grad| max %.3e, diff max %.3e at %s, diff rel l2 %.3e" % (
        i, tuple(a.shape), a.abs().max().item(), d.max().item(), tuple(int(v) for v in (d == d.max()).nonzero()[0]),
        (a - b).norm().item() / a.norm().item()))
    d2 = d.reshape(-1, d.shape[-2], d.shape[-1]).sum(0)
    rows = d2.sum(1)
    cols = d2.sum(0)
    print("   diff mass by row block of 44: %s" % [round(float(v), 6) for v in rows.reshape(-1, 44).sum(1)[:10]])
    print("   diff mass by col block of 128: %s" % [round(float(v), 6) for v in cols.reshape(-1, 128).sum(1)])
    frac = (d > 1e-3 * a.abs().max()).float().mean().item()
    print("   entries with diff > 1e-3 of max|grad|: %.4f%%" % (100 * frac))
