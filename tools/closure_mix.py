#!/usr/bin/env python3
"""Kernel mix of ONE eagerly launched closure (torch.profiler device timestamps).  usage: closure_mix.py NET [HxW] [top]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    acc = {}
    for ev in prof.events():
        if ev.device_type == DeviceType.CUDA:
            a = acc.setdefault(ev.name[:120], [0, 0.0])
            a[0] += 1
            a[1] += ev.time_range.elapsed_us()
    tot = sum(v[1] for v in acc.values())
    print("%s %dx%d closure: device time %.2f ms, %d launches" % (net, h, w, tot / 1e3, sum(v[0] for v in acc.values())))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-120s %4d %9.1f us %6.1f avg %5.2f%%" % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))


if __name__ == "__main__":
    main()
