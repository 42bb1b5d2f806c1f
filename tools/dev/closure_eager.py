#!/usr/bin/env python3
"""N eagerly launched closures of one net (a target for `rocprofv3 --kernel-trace`): closure_eager.py [NET] [HxW] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
for _ in range(n):
    st.optimizer.zero_grad()
    st._closure_body()
torch.cuda.synchronize()
