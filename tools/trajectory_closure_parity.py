#!/usr/bin/env python3
"""Closure-level parity ALONG a real attack trajectory (robust to the optimiser's knife-edge decisions).

The CPU port runs `--steps` attack steps and records the variables at every closure evaluation (10 per step, the
overshoot points of the fixed-step L-BFGS included); the GPU then evaluates its closure at exactly those points.  Compared
per point: loss (relative) and d(loss)/d(variables) (relative L2).  End-of-attack metrics of two runs can sit on
different branches of the optimiser's hard thresholds; the loss and gradient at a GIVEN point
cannot.

    python tools/trajectory_closure_parity.py [--net RAFT] [--size 436x1024] [--steps 3] [--threads 16] [--seed 0]
                                              [--box change_of_variables] [--joint] [--out FILE.json]
Exit code 0 iff every point is inside --loss-tol / --grad-tol."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="RAFT")
    ap.add_argument("--size", default="436x1024")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--box", default="change_of_variables", choices=["change_of_variables", "clipping"])
    ap.add_argument("--joint", action="store_true")
    ap.add_argument("--loss-tol", type=float, default=1e-5)
    ap.add_argument("--grad-tol", type=float, default=1e-2, help="relative L2, as test_closure_at_baseline_size_vs_cpu_port")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from oracle import ops as oracle_ops
    from pcfa_amd import ops
    h, w = (int(v) for v in a.size.split("x"))
    kw = dict(boxconstraint=a.box, joint=a.joint)
    torch.set_num_threads(a.threads)
    t0 = time.perf_counter()
    points = []
    with ops.override_for_testing(oracle_ops):
        cst = bench.AttackStepper(a.net, h, w, torch.device("cpu"), seed=a.seed, **kw)
        inner = cst.closure

        def recording():
            x = [p.detach().clone() for p in cst.params]
            loss = inner()
            points.append((x, float(loss), torch.cat([p.grad.detach().flatten() for p in cst.params]).clone()))
            return loss
        cst.closure = recording
        for k in range(a.steps):
            cst.step()
            print("port step %d/%d: %d points (%.0f s)" % (k + 1, a.steps, len(points), time.perf_counter() - t0),
                  file=sys.stderr, flush=True)
    gst = bench.AttackStepper(a.net, h, w, torch.device("cuda", 0), seed=a.seed, **kw)
    rows, ok = [], True
    for i, (x, loss_c, grad_c) in enumerate(points):
        with torch.no_grad():
            for p, v in zip(gst.params, x):
                p.copy_(v.to(p.device))
        gst.optimizer.zero_grad()
        loss_g = float(gst._closure_body())
        grad_g = torch.cat([p.grad.detach().flatten() for p in gst.params]).cpu()
        lr = abs(loss_g - loss_c) / abs(loss_c)
        gr = float((grad_g - grad_c).norm() / grad_c.norm())
        rows.append({"point": i, "loss_port": loss_c, "loss_gpu": loss_g, "loss_rel": lr, "grad_rel_l2": gr})
        ok = ok and lr <= a.loss_tol and gr <= a.grad_tol
    out = {"what": "loss and gradient of the GPU closure at every iterate of a %d-step CPU-port attack (%d points), %s %dx%d, "
                   "%s%s, synthetic pair %d" % (a.steps, len(points), a.net, h, w, a.box, ", joint" if a.joint else "", a.seed),
           "ok": ok, "loss_tol": a.loss_tol, "grad_tol": a.grad_tol,
           "max_loss_rel": max(r["loss_rel"] for r in rows), "max_grad_rel_l2": max(r["grad_rel_l2"] for r in rows),
           "loss_range": [min(r["loss_port"] for r in rows), max(r["loss_port"] for r in rows)], "points": rows}
    txt = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
