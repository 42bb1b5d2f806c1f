#!/usr/bin/env python3
"""Debug aid: FlowNet2 closure eager vs hipGraph replay (loss / gradient finiteness)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def stats(tag, st, loss):
    g1, g2 = st.nw1.grad, st.nw2.grad
    print("%-28s loss %.6g  |g1| %.4g  |g2| %.4g  finite %s %s" % (
        tag, float(loss), float(g1.norm()), float(g2.norm()), bool(torch.isfinite(g1).all()),
        bool(torch.isfinite(g2).all())), flush=True)


def main():
    h, w = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x128").split("x"))
    st = bench.AttackStepper("FlowNet2", h, w, torch.device("cuda", 0), seed=0)
    for i in range(3):
        st.nw1.grad = st.nw2.grad = None
        loss = st._closure_body()
        stats("eager closure %d" % i, st, loss.detach())
        del loss
    st.nw1.grad = st.nw2.grad = None
    st.enable_graph()
    for i in range(3):
        loss = st.graphed()
        stats("graph replay %d" % i, st, loss)
    with torch.no_grad():
        f = st.predict()
    print("predict finite", bool(torch.isfinite(f).all()), float(f.abs().max()))
    d1, d2, f = st.repredict()
    print("repredict finite", bool(torch.isfinite(f).all()), float(f.abs().max()))
    r = st.step()
    print("step ->", r, "closures", st.closures)


if __name__ == "__main__":
    main()
