#!/usr/bin/env python3
"""Which host operations launch the library's element-wise / copy kernels inside one eager closure (the glue between the
hand-written kernels): aten ops with device time, grouped by name, input shapes and the innermost frame under the repo.
usage: glue_kernels.py [NET] [HxW] [joint]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "PWCNet"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "375x1242").split("x"))
    joint = len(sys.argv) > 3
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0, boxconstraint="clipping" if joint else
                             "change_of_variables", joint=joint, use_graph=False)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    rows = collections.OrderedDict()
    for e in prof.events():
        if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
            continue
        frame = next((f for f in (e.stack or []) if ROOT in f and "tools/dev" not in f and "bench.py" not in f), "(autograd engine)")
        frame = frame.replace(ROOT + "/", "")
        key = (e.name, str(e.input_shapes)[:70], frame[:90])
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += e.self_device_time_total
    tot = sum(v[1] for v in rows.values())
    print("%s %dx%d: %d library operations with device time, %.1f us per closure" % (net, h, w, sum(v[0] for v in rows.values()), tot))
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%7.1f us %3d x  %-18s %-70s %s" % (v[1], v[0], k[0], k[1], k[2]))


if __name__ == "__main__":
    main()
