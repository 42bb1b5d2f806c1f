#!/usr/bin/env python3
"""Where the small elementwise launches of one closure come from: for every CPU op of the given kinds (default
aten::add / add_ / fill_ / zero_ / copy_ / mul / neg ...) the enclosing autograd node (or forward scope) and the input
shapes, counted.  usage: small_ops.py [NET] [HxW]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
    leaf = collections.Counter()
    for e in evs:
        if not e.name.startswith("aten::") or not e.kernels:
            continue
        if any(c.name.startswith("aten::") and c.kernels for c in e.cpu_children):
            continue   # count the innermost aten op that launched
        p, chain = e.cpu_parent, []
        while p is not None:
            if not p.name.startswith("aten::"):
                chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:")[:48])
            p = p.cpu_parent
        dev_us = sum(k.duration for k in e.kernels)
        shapes = str([s for s in (e.input_shapes or []) if s])[:70]
        leaf[(e.name, " < ".join(chain[:2]), shapes)] += 1
        leaf[("~us", e.name, "")] += dev_us
    tot = collections.Counter()
    for (n, c, s), v in leaf.items():
        if n != "~us":
            tot[n] += v
    for n, v in tot.most_common(25):
        print("%-28s %4d launching calls, %8.1f us device" % (n, v, leaf[("~us", n, "")]))
    print()
    kinds = ("aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::copy_", "aten::mul", "aten::neg", "aten::sub",
             "aten::div", "aten::sum", "aten::cat", "aten::clone", "aten::_to_copy")
    for (n, c, s), v in sorted(leaf.items(), key=lambda kv: (kv[0][0], -kv[1])):
        if n in kinds:
            print("%-14s x%3d  %-70s %s" % (n, v, c, s))


if __name__ == "__main__":
    main()
