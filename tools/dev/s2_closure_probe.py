#!/usr/bin/env python3
"""Eager RAFT closure at 436x1024 at several points: library stride-2 layers vs conv_s2 (loss and gradient)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd import hip_ops  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=0)
gen = torch.Generator().manual_seed(3)
base = [p.detach().clone() for p in st.params]
CONFIGS = [("lib", (False, False, False)), ("fwd", (True, False, False)), ("fwd+bwd", (True, True, False)),
           ("all", (True, True, True))]
for sigma in (0.0, 1e-3):
    noise = [torch.randn(p.shape, generator=gen).to(p.device) for p in st.params]
    res = {}
    for name, (a_, b_, c_) in CONFIGS:
        raft.CONV_S2, hip_ops.CONV_S2_BWD, raft.FUSED_DOWNSAMPLE = a_, b_, c_
        with torch.no_grad():
            for p, b, n in zip(st.params, base, noise):
                p.copy_(b + sigma * n)
        st.optimizer.zero_grad()
        loss = st._closure_body()
        res[name] = (float(loss), torch.cat([p.grad.flatten() for p in st.params]).clone())
    for name, _ in CONFIGS[1:]:
        dl = abs(res["lib"][0] - res[name][0]) / abs(res["lib"][0])
        dg = (res["lib"][1] - res[name][1]).norm().item() / res["lib"][1].norm().item()
        print("sigma %g %-8s: loss %.7g vs lib %.7g (rel %.2e)  grad rel diff %.3e" % (sigma, name, res[name][0], res["lib"][0], dl, dg), flush=True)
    # the library against itself (run-to-run) for scale
    raft.CONV_S2 = False
    st.optimizer.zero_grad()
    st._closure_body()
    g2 = torch.cat([p.grad.flatten() for p in st.params])
    print("sigma %g lib rerun: grad rel diff %.3e" % (sigma, (res["lib"][1] - g2).norm().item() / g2.norm().item()), flush=True)
