#!/usr/bin/env python3
"""Where does one attack step's wall time go?  Times, at the bench shape: the captured closure replayed back to back
(pure GPU time of forward + loss + backward), the captured re-prediction forward, and whole steps -- the difference
is L-BFGS's own vector math and host synchronisations.  `step_breakdown.py RAFT torch` swaps in torch.optim.LBFGS."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    if len(sys.argv) > 2 and sys.argv[2] == "torch":  # A/B: the stock optimiser
        from pcfa_amd import hip_ops
        hip_ops.LBFGS = torch.optim.LBFGS
    dev = torch.device("cuda", 0)
    st = bench.AttackStepper(net, 436, 1024, dev, seed=0)
    st.step()
    st.enable_graph()
    st.step()
    torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_closure = timed(st.graphed, 20)
    t_fwd = timed(st.repredict, 20)
    hist = []
    for _ in range(int(os.environ.get("STEPS", "12"))):   # history reaches 100 pairs after 10 steps
        os_ = st.optimizer.state[st.optimizer._params[0]]
        n_old = st.optimizer.history_count() if hasattr(st.optimizer, "history_count") else len(os_.get("old_dirs", []))
        c0 = st.closures
        t = timed(st.step, 1)
        hist.append((n_old, st.closures - c0, t))
    print("%s: closure graph replay %.2f ms   re-prediction forward replay %.2f ms" % (net, t_closure, t_fwd))
    for n_old, nc, t in hist:
        rest = t - nc * t_closure - t_fwd
        print("step with L-BFGS history %3d: %7.2f ms = %d closures x %.2f + forward %.2f + %.2f ms other "
              "(%.2f ms per L-BFGS iteration: two-loop recursion, line-search bookkeeping, host syncs)" %
              (n_old, t, nc, t_closure, t_fwd, rest, rest / max(nc, 1)))


if __name__ == "__main__":
    main()
