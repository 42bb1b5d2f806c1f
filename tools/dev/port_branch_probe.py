#!/usr/bin/env python3
"""Which branch does the CPU port take at the third closure of the first attack step (RAFT 436x1024)?  The loss of the
first four closures for several thread counts (different reduction orders inside the CPU convolutions)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ops as oracle_ops  # noqa: E402
from pcfa_amd import ops  # noqa: E402


class Stop(Exception):
    pass


NOMKLDNN = len(sys.argv) > 2 and sys.argv[2] == "nomkldnn"   # torch's native CPU convolution instead of oneDNN's
torch.backends.mkldnn.enabled = not NOMKLDNN
for threads in [int(v) for v in sys.argv[1].split(",")]:
    torch.set_num_threads(threads)
    with ops.override_for_testing(oracle_ops):
        st = bench.AttackStepper("RAFT", 436, 1024, torch.device("cpu"), seed=0)
        losses = []
        orig = st.closure

        def spy():
            l = orig()
            losses.append(float(l))
            if len(losses) >= 4:
                raise Stop
            return l

        st.closure = spy
        try:
            st.step()
        except Stop:
            pass
    print("port threads %2d %s closures: %s" % (threads, "native conv" if NOMKLDNN else "oneDNN", ["%.6f" % v for v in losses]),
          flush=True)
