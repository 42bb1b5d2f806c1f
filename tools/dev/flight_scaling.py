#!/usr/bin/env python3
"""pair-steps/s of n pairs in flight on one GPU (attack_PCFA.PairsInFlight; hipGraph closures, STEPS timed steps after 1
warm-up).  flight_scaling.py [NET] [HxW] [joint]; FLIGHT_N=1,2,3,4 selects the lane counts, FLIGHT_STEPS the steps (4)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
joint = len(sys.argv) > 3
box = "clipping" if joint else "change_of_variables"
steps = int(os.environ.get("FLIGHT_STEPS", "4"))
if os.environ.get("FLIGHT_WATCHDOG_S"):     # a hung run dumps every thread's stack and exits
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["FLIGHT_WATCHDOG_S"]), exit=True)
dev = torch.device("cuda", 0)
model = bench.load_model(net, dev, not joint)
for n in tuple(int(v) for v in os.environ.get('FLIGHT_N', '1,2,3,4').split(',')):
    model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
    f = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper(net, h, w, dev, 600 + k, boxconstraint=box, joint=joint,
                                                                use_graph=True, model=model), n, dev)
    f.run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s %dx%d pairs in flight %d: %.2f pair-steps/s (%.1f ms per step per pair)" % (net, h, w, n, n * steps / dt, 1e3 * dt / steps),
          flush=True)
    del f
