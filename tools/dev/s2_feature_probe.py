#!/usr/bin/env python3
"""Feature / context encoder outputs at 440x1024: CPU port vs GPU (library stride-2 layers) vs GPU (conv_s2)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ops as oracle_ops  # noqa: E402
from pcfa_amd import hip_ops, ops  # noqa: E402
from pcfa_amd.helper_functions import datasets  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

torch.set_num_threads(16)
dev = torch.device("cuda", 0)
gm = bench.load_model("RAFT", dev, True)
with ops.override_for_testing(oracle_ops):
    cm = bench.load_model("RAFT", torch.device("cpu"), True)


def find(m):
    for mod in m.modules():
        if hasattr(mod, "fnet") and hasattr(mod, "cnet"):
            return mod
    raise RuntimeError


gr, cr = find(gm), find(cm)
i1, i2, _ = datasets.synthetic_pair(0, 436, 1024)
x1 = torch.nn.functional.pad(i1[None], (0, 0, 2, 2), mode="replicate")
x2 = torch.nn.functional.pad(i2[None], (0, 0, 2, 2), mode="replicate")
a1, a2 = (2 * (x1 / 255.0) - 1.0).contiguous(), (2 * (x2 / 255.0) - 1.0).contiguous()
with torch.no_grad():
    with ops.override_for_testing(oracle_ops):
        cf1, cf2 = cr.fnet([a1, a2])
        cc = cr.cnet(a1)
    outs = {}
    for name, on in (("lib", False), ("s2", True)):
        raft.CONV_S2 = on
        f1, f2 = gr.fnet([a1.to(dev), a2.to(dev)])
        c = gr.cnet(a1.to(dev))
        outs[name] = (f1.cpu(), f2.cpu(), c.cpu())
for name in ("lib", "s2"):
    f1, f2, c = outs[name]
    print("%s vs port: fmap1 rel l2 %.3e max %.3e | fmap2 %.3e | cnet %.3e max %.3e" % (
        name, (f1 - cf1).norm() / cf1.norm(), (f1 - cf1).abs().max(), (f2 - cf2).norm() / cf2.norm(),
        (c - cc).norm() / cc.norm(), (c - cc).abs().max()))
print("s2 vs lib : fmap1 rel l2 %.3e | cnet %.3e" % ((outs["s2"][0] - outs["lib"][0]).norm() / outs["lib"][0].norm(),
                                                     (outs["s2"][2] - outs["lib"][2]).norm() / outs["lib"][2].norm()))
