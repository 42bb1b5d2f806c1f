#!/usr/bin/env python3
"""Which torch (aten) operators still launch kernels inside one RAFT closure, with input shapes and python call sites:
aten_ops_probe.py [NET]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "RAFT"
    st = bench.AttackStepper(net, 436, 1024, torch.device("cuda", 0), seed=0)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=0 if "--flat" in sys.argv else 6):
        dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if dt > 0 and ev.key.startswith("aten::"):
            stack = [s for s in ev.stack if "pcfa_amd" in s or "bench.py" in s][:3]
            rows.append((dt, ev.count, ev.key, str(ev.input_shapes)[:90], " <- ".join(s.split("/")[-1] for s in stack)))
    for dt, n, key, shp, stack in sorted(rows, reverse=True)[:60]:
        print("%8.1f us %3d  %-28s %-90s %s" % (dt, n, key, shp, stack))


if __name__ == "__main__":
    main()
