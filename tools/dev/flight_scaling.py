#!/usr/bin/env python3
"""pair-steps/s of n RAFT pairs in flight on one GPU, n = 1..4 (attack_PCFA.PairsInFlight; 436x1024, 4 steps after 1 warm-up)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.load_model("RAFT", dev, True)
for n in tuple(int(v) for v in os.environ.get('FLIGHT_N', '1,2,3,4').split(',')):
    model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
    f = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", 436, 1024, dev, 600 + k, use_graph=True, model=model), n, dev)
    f.run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.run(4)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pairs in flight %d: %.2f pair-steps/s (%.1f ms per step per pair)" % (n, n * 4 / dt, 1e3 * dt / 4), flush=True)
    del f
