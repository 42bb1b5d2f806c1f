#!/usr/bin/env python3
"""The two in-flight bit-identity tests of tests/test_gpu_parity.py, N times each in ONE process (they failed once in a full
suite run and never alone): flight_tests_repeat.py [N]"""
import os
import pathlib
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_gpu_parity as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = {"solo": 0, "attack_l2": 0}
for it in range(N):
    try:
        T.test_pairs_in_flight_bit_identical_to_solo()
    except AssertionError as e:
        bad["solo"] += 1
        print("iteration %d: in flight vs solo: %s" % (it, str(e).splitlines()[0][:300]), flush=True)
    with tempfile.TemporaryDirectory() as d:
        try:
            T.test_attack_l2_pairs_in_flight_equals_sequential("RAFT", False, "change_of_variables", pathlib.Path(d))
        except AssertionError as e:
            bad["attack_l2"] += 1
            print("iteration %d: attack_l2 in flight vs sequential: %s" % (it, str(e).splitlines()[0][:300]), flush=True)
    print("iteration %d done" % it, file=sys.stderr, flush=True)
print("iterations: %d, failures: %r" % (N, bad), flush=True)
