#!/usr/bin/env python3
"""The values test_pairs_in_flight_bit_identical_to_solo compares, printed (its model, seeds and sizes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402
from tests import closure_util  # noqa: E402

dev = torch.device("cuda", 0)
model = closure_util.load_model("RAFT", True, dev)
flight = attack_PCFA.PairsInFlight(lambda k: bench.AttackStepper("RAFT", 128, 160, dev, (11, 12)[k], use_graph=True, model=model), 2, dev)
last = flight.run(2)
print("flight:", [tuple(v) for v in last])
for seed in (11, 12):
    model._pcfa_pair_graphs.clear()
    solo = bench.AttackStepper("RAFT", 128, 160, dev, seed, use_graph=True, model=model)
    for _ in range(2):
        v = solo.step()
    print("solo %d:" % seed, tuple(v))
