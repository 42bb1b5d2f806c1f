#!/usr/bin/env python3
"""Is a two-lane flight bit-identical to the solo runs EVERY time?  flight_repeat.py [N] [HxW] -- N flights of pairs (11, 12)
against one solo reference each; prints the lanes that differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128x160").split("x"))
STEPS = 2
dev = torch.device("cuda", 0)
model = bench.load_model("RAFT", dev, True)


def solo(seed):
    model._pcfa_pair_graphs.clear() if hasattr(model, "_pcfa_pair_graphs") else None
    st = bench.AttackStepper("RAFT", H, W, dev, seed, use_graph=True, model=model)
    for _ in range(STEPS):
        last = st.step()
    return tuple(last), st.delta1.clone(), st.flow_pred.clone()


refs = [solo(11), solo(12)]
again = [solo(11), solo(12)]
print("solo twice:", all(a[0] == b[0] and torch.equal(a[1], b[1]) for a, b in zip(refs, again)), flush=True)
bad = 0
if os.environ.get("FLIGHT_SOLO_ONLY") == "1":      # the same number of fresh captures, one pair at a time
    for it in range(N):
        for k, seed in enumerate((11, 12)):
            got = solo(seed)
            if not (got[0] == refs[k][0] and torch.equal(got[1], refs[k][1])):
                bad += 1
                print("solo run %d seed %d DIFFERENT: %r vs %r" % (it, seed, got[0], refs[k][0]), flush=True)
    print("solo runs: %d x 2, differed: %d" % (N, bad), flush=True)
    sys.exit(0)
for it in range(N):
    model._pcfa_pair_graphs.clear()
    flight = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", H, W, dev, (11, 12)[k], use_graph=True, model=model), 2, dev)
    last = flight.run(STEPS)
    for k in (0, 1):
        same = tuple(last[k]) == refs[k][0] and torch.equal(flight.attacks[k].delta1, refs[k][1])
        if not same:
            bad += 1
            print("flight %d lane %d DIFFERENT: %r vs %r" % (it, k, tuple(last[k]), refs[k][0]), flush=True)
    del flight
print("flights: %d, lanes that differed: %d" % (N, bad), flush=True)
