#!/usr/bin/env python3
"""Is the first attack of a process different from the second one on the same pair?  usage: firstrun_probe.py [gram|two_loop] [graph|eager]"""
import functools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import closure_util  # noqa: E402
from tests.util import rel_l2  # noqa: E402
from pcfa_amd import attack_PCFA, hip_ops, lbfgs  # noqa: E402
from pcfa_amd.helper_functions import datasets  # noqa: E402


def main():
    direction = sys.argv[1] if len(sys.argv) > 1 else "gram"
    graph = (sys.argv[2] if len(sys.argv) > 2 else "graph") == "graph"
    hip_ops.LBFGS = functools.partial(lbfgs.LBFGS, direction=direction)
    dev = torch.device("cuda:0")
    args = closure_util.cli_args(net="RAFT", steps=2)
    mu = attack_PCFA.default_mu(args)
    model = closure_util.load_model("RAFT", True, dev)
    i1, i2, _ = datasets.synthetic_pair(0, 128, 160)
    outs = []
    for rep in range(3):
        st = attack_PCFA.PairAttack(model, i1[None], i2[None], None, 0, attack_PCFA.EPS_BOX, dev, False, mu, args,
                                    use_graph=graph, reuse_graphs=False)
        losses = []
        orig = st.closure

        def wrapped():
            l = orig()
            losses.append(float(l))
            return l
        st.closure = wrapped
        hist = [st.step() for _ in range(2)]
        outs.append((st.delta1.detach().clone(), hist, losses))
        print("rep %d: %s" % (rep, hist))
        print("   losses:", " ".join("%.6f" % l for l in losses))
    print("rel rep0-rep1 %.3e   rep1-rep2 %.3e" % (rel_l2(outs[0][0], outs[1][0]), rel_l2(outs[1][0], outs[2][0])))


if __name__ == "__main__":
    main()
