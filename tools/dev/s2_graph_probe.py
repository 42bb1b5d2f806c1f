#!/usr/bin/env python3
"""Graph-replayed closure vs eager closure with conv_s2 on (and off), same point."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd import hip_ops  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

for name, on in (("lib", False), ("s2", True)):
    raft.CONV_S2 = on
    st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=0)
    st.optimizer.zero_grad()
    le = float(st._closure_body())
    ge = torch.cat([p.grad.flatten() for p in st.params]).clone()
    st.enable_graph()
    for rep in range(3):
        st.optimizer.zero_grad()
        lg = float(st.closure())
        gg = torch.cat([p.grad.flatten() for p in st.params]).clone()
        print("%s replay %d: loss eager %.7g graph %.7g, grad rel diff %.3e" % (name, rep, le, lg, (gg - ge).norm().item() / ge.norm().item()), flush=True)
