#!/usr/bin/env python3
"""Per-pair difference between PairAttack with graph reuse and with fresh captures (and fresh vs fresh = the noise)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import closure_util  # noqa: E402
from tests.util import rel_l2  # noqa: E402
from pcfa_amd import attack_PCFA  # noqa: E402
from pcfa_amd.helper_functions import datasets  # noqa: E402


def run(reuse, steps):
    dev = torch.device("cuda:0")
    args = closure_util.cli_args(net="RAFT", steps=steps)
    mu = attack_PCFA.default_mu(args)
    model = closure_util.load_model("RAFT", True, dev)
    if hasattr(model, "_pcfa_pair_graphs"):
        model._pcfa_pair_graphs.clear()
    out = []
    for seed, (h, w) in ((0, (128, 160)), (1, (128, 160)), (2, (136, 168)), (3, (128, 160))):
        i1, i2, _ = datasets.synthetic_pair(seed, h, w)
        st = attack_PCFA.PairAttack(model, i1[None], i2[None], None, seed, attack_PCFA.EPS_BOX, dev, False, mu, args,
                                    use_graph=True, reuse_graphs=reuse)
        hist = []
        for _ in range(steps):
            hist.append(st.step())
        out.append((st.graphs_reused, st.delta1.detach().clone(), hist))
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    a, b, c = run(True, steps), run(False, steps), run(False, steps)
    for i in range(4):
        print("pair %d reused=%s  reuse-vs-fresh %.3e   fresh-vs-fresh %.3e" % (i, a[i][0], rel_l2(a[i][1], b[i][1]),
                                                                              rel_l2(b[i][1], c[i][1])))
        print("    reuse ", a[i][2][-1], "\n    fresh ", b[i][2][-1], "\n    fresh2", c[i][2][-1])


if __name__ == "__main__":
    main()
