#!/usr/bin/env python3
"""Setup time of later pairs of the same shape (graph reuse), with a breakdown."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd.helper_functions import datasets  # noqa: E402

dev = torch.device("cuda", 0)
st = bench.AttackStepper("RAFT", 436, 1024, dev, seed=0, use_graph=True)
st.step()
torch.cuda.synchronize()
for k in range(4):
    t0 = time.perf_counter()
    datasets.synthetic_pair(1000 + k, 436, 1024)
    t1 = time.perf_counter()
    st2 = bench.AttackStepper("RAFT", 436, 1024, dev, seed=1000 + k, use_graph=True, model=st.model)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("pair %d: synthetic_pair on the host %.3f s, AttackStepper (incl. its own synthetic_pair) %.3f s, reused %s"
          % (k, t1 - t0, t2 - t1, st2.graphs_reused), flush=True)
    del st2
