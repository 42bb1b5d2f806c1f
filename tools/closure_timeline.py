#!/usr/bin/env python3
"""Ordered kernel timeline of ONE hipGraph replay of a closure (device timestamps from the HIP activity tracer): every launch
with its start offset, duration and the idle gap before it.  usage: closure_timeline.py [NET] [HxW] [first] [count]
Prints launches [first, first + count) of the replay (default: all) and the totals (busy, idle, span)."""
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "RAFT"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "436x1024").split("x"))
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0)
    st.step()
    st.enable_graph()
    for _ in range(3):
        st.graphed()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        st.graphed()
        torch.cuda.synchronize()
    evs = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
    t0, prev_end, busy = evs[0].time_range.start, evs[0].time_range.start, 0.0
    for i, e in enumerate(evs):
        gap = e.time_range.start - prev_end
        dur = e.time_range.elapsed_us()
        busy += dur
        if first <= i < first + count:
            print("%4d  +%9.1f us  gap %5.1f  dur %7.1f  %s" % (i, e.time_range.start - t0, gap, dur, short(e.name)))
        prev_end = max(prev_end, e.time_range.end)
    span = prev_end - t0
    print("launches %d  busy %.1f us  span %.1f us  idle %.1f us (%.1f %%)" % (len(evs), busy, span, span - busy,
                                                                              100 * (span - busy) / span))


if __name__ == "__main__":
    main()
