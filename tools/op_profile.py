#!/usr/bin/env python3
"""Which aten ops launch the elementwise / copy / fill kernels of one RAFT closure?  torch.profiler over ONE eagerly
launched closure (fwd + loss + bwd) at the bench shape, grouped by op and input shape, sorted by device time."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    st = bench.AttackStepper("RAFT", 436, 1024, dev, seed=0)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True)
    rows = sorted(ka, key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print("total self device time %.2f ms" % (tot / 1e3))
    print("%-44s %6s %10s %6s  %s" % ("op", "calls", "dev_us", "pct", "input shapes"))
    for e in rows[:160]:
        if e.self_device_time_total <= 0:
            continue
        print("%-44s %6d %10.1f %6.2f  %s" % (e.key[:44], e.count, e.self_device_time_total,
                                              100.0 * e.self_device_time_total / tot, str(e.input_shapes)[:150]))


if __name__ == "__main__":
    main()
