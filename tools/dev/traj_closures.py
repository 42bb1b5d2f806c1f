#!/usr/bin/env python3
"""Loss of every closure of the first two attack steps (RAFT 436x1024, graphs on).  argv: s2 on|off"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

raft.CONV_S2 = sys.argv[1] == "on"
st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=0, use_graph=True)
losses = []
orig = st.closure


def spy():
    l = orig()
    losses.append(float(l))
    return l


st.closure = spy
if hasattr(st, "optimizer"):
    pass
for k in range(2):
    m = st.step()
    print("step", k, [round(float(v), 5) for v in m], flush=True)
print("conv_s2 %s algo %s closures:" % (sys.argv[1], os.environ.get("PCFA_CONV3X3_ALGO", "policy")), ["%.6f" % v for v in losses])
