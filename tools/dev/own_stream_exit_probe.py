#!/usr/bin/env python3
"""Which use of an ExternalStream makes the interpreter crash at exit?  own_stream_exit_probe.py MODE"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import ops  # noqa: E402

mode = sys.argv[1]
dev = torch.device("cuda", 0)
x = torch.ones(1 << 20, device=dev)
s = ops.core.own_stream(dev, 0)
if mode == "eager":
    with torch.cuda.stream(s):
        y = (x * 2).sum()
    s.synchronize()
    print(float(y))
elif mode == "wait":
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = (x * 2).sum()
    torch.cuda.current_stream().wait_stream(s)
    print(float(y))
elif mode in ("graph", "graph_del"):
    g = torch.cuda.CUDAGraph()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = (x * 2).sum()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        y = (x * 2).sum()
    g.replay()
    torch.cuda.synchronize()
    print(float(y))
    if mode == "graph_del":
        del g, y
        torch.cuda.synchronize()
elif mode == "pool_graph":
    s = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        y = (x * 2).sum()
    g.replay()
    torch.cuda.synchronize()
    print(float(y))
print("done", mode, flush=True)
