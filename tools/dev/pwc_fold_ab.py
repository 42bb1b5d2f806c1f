#!/usr/bin/env python3
"""PWC-Net closure (375x1242, joint, clipping; hipGraph replay) with Config.pwc_fold_glue on and off in ONE process on one
box: device time per closure from events around 50 replays, launches per closure from the graph's kernel count.
usage: pwc_fold_ab.py"""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd import config  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for name, conf in (("pwc_fold_glue=True (default)", config.DEFAULT),
                       ("pwc_fold_glue=False", dataclasses.replace(config.DEFAULT, pwc_fold_glue=False))):
        st = bench.AttackStepper("PWCNet", 375, 1242, dev, seed=0, boxconstraint="clipping", joint=True, use_graph=True,
                                 config=conf)
        for _ in range(5):
            st.closure()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            st.closure()
        b.record()
        torch.cuda.synchronize()
        print("%-32s %.3f ms per closure (graphed: %s)" % (name, a.elapsed_time(b) / 50, st.graphed is not None))
        del st


if __name__ == "__main__":
    main()
