#!/usr/bin/env python3
"""d(loss)/d(variables) of one RAFT closure at 436x1024, sigma = 0 and small: CPU port vs GPU with the library stride-2
layers vs GPU with conv_s2."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ops as oracle_ops  # noqa: E402
from pcfa_amd import hip_ops, ops  # noqa: E402
from pcfa_amd.nets import raft  # noqa: E402

torch.set_num_threads(16)
gst = bench.AttackStepper("RAFT", 436, 1024, torch.device("cuda", 0), seed=0)
with ops.override_for_testing(oracle_ops):
    cst = bench.AttackStepper("RAFT", 436, 1024, torch.device("cpu"), seed=0)
gen = torch.Generator().manual_seed(3)
gbase = [p.detach().clone() for p in gst.params]
cbase = [p.detach().clone() for p in cst.params]
for sigma in (0.0, 2e-3):
    noise = [torch.randn(p.shape, generator=gen) for p in cst.params]
    with torch.no_grad():
        for p, b, n in zip(cst.params, cbase, noise):
            p.copy_(b + sigma * n)
    with ops.override_for_testing(oracle_ops):
        cst.optimizer.zero_grad()
        lc = float(cst._closure_body())
    gc = torch.cat([p.grad.flatten() for p in cst.params])
    for name, on in (("lib", False), ("s2", True)):
        raft.CONV_S2 = on
        with torch.no_grad():
            for p, b, n in zip(gst.params, gbase, noise):
                p.copy_(b + sigma * n.to(p.device))
        gst.optimizer.zero_grad()
        lg = float(gst._closure_body())
        gg = torch.cat([p.grad.flatten() for p in gst.params]).cpu()
        print("sigma %g gpu[%s] vs port: loss %.7g / %.7g, grad rel l2 %.3e" % (sigma, name, lg, lc, (gg - gc).norm().item() / gc.norm().item()), flush=True)
