#!/usr/bin/env python3
"""PCFA_SHARED_FORWARD experiment (graphed.SplitGraphedClosure): (1) two attack steps at 128x160 with and without the
shared forward -- per-step metrics and final perturbations; (2) step time at the bench shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)


def run(share, h, w, steps, model=None):
    st = bench.AttackStepper("RAFT", h, w, dev, seed=0, model=model)
    st.enable_graph(share_forward=share)
    assert st.graphed is not None, "capture failed"
    out = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out.append(st.step())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return st, out, dt


model = bench.load_model("RAFT", dev, True)
a, ma, _ = run(False, 128, 160, 2, model)
b, mb, _ = run(True, 128, 160, 2, model)
print("metrics default:", ma)
print("metrics shared :", mb)
rel = lambda x, y: float((x - y).norm() / y.norm())  # noqa: E731
print("final delta1 rel diff %.3e  delta2 %.3e  flow %.3e" % (rel(b.delta1, a.delta1), rel(b.delta2, a.delta2),
                                                               rel(b.flow_pred, a.flow_pred)))
print("forwards shared:", b.graphed.forwards_shared, "of", b.graphed.replays, "closure evaluations")
for share in (False, True, False, True):
    st, _, _ = run(share, 436, 1024, 1, model)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        st.step()
    torch.cuda.synchronize()
    print("436x1024 shared=%s: %.2f ms per step" % (share, (time.perf_counter() - t0) / 3 * 1e3))
