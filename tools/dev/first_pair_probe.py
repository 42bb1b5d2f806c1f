#!/usr/bin/env python3
"""The first curvature pair of torch.optim.LBFGS on the bench problem (RAFT 436x1024): y.s against the optimiser's hard
acceptance threshold 1e-10, with the library stride-2 layers and with conv_s2."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

SEEDS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
TARGET = sys.argv[2] if len(sys.argv) > 2 else "zero"
for seed, name, on in [(sd, nm, o) for sd in SEEDS for nm, o in (("lib", False), ("s2", True))]:
    raft.CONV_S2 = on
    st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=seed, target=TARGET)
    lr = float(st.optimizer.param_groups[0]["lr"])
    base = [p.detach().clone() for p in st.params]
    st.optimizer.zero_grad()
    l0 = float(st._closure_body())
    g0 = torch.cat([p.grad.flatten() for p in st.params]).double().clone()
    t = min(1.0, 1.0 / g0.abs().sum().item()) * lr
    with torch.no_grad():
        off = 0
        for p in st.params:
            n = p.numel()
            p.add_((-t * g0[off:off + n]).float().view_as(p))
            off += n
    st.optimizer.zero_grad()
    l1 = float(st._closure_body())
    g1 = torch.cat([p.grad.flatten() for p in st.params]).double().clone()
    y, s = g1 - g0, -t * g0
    ys, yy = float(y @ s), float(y @ y)
    print("seed %d %s: lr %g t %.4g loss %.7f -> %.7f |g0| %.4e |y|/|g0| %.3e  y.s = %.4e (threshold 1e-10)  gamma = ys/yy = %.4g"
          % (seed, name, lr, t, l0, l1, g0.norm().item(), y.norm().item() / g0.norm().item(), ys, ys / yy), flush=True)
