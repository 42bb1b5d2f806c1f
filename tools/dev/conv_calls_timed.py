#!/usr/bin/env python3
"""Every pcfa_conv3x3_run call of one eager closure with its shape and device time (profiler kernel events matched to the
calls in launch order; a call whose input channels are split over workgroups owns two kernels: partial + finish), sorted by total time.
usage: conv_calls_timed.py [NET] [HxW] [joint]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from pcfa_amd.ops import core  # noqa: E402


def main():
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    net = sys.argv[1] if len(sys.argv) > 1 else "PWCNet"
    h, w = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "375x1242").split("x"))
    joint = len(sys.argv) > 3
    st = bench.AttackStepper(net, h, w, torch.device("cuda", 0), seed=0, boxconstraint="clipping" if joint else
                             "change_of_variables", joint=joint)
    for _ in range(2):
        st.optimizer.zero_grad()
        st._closure_body()
    torch.cuda.synchronize()
    calls = []
    orig = core._WORK_TABLE["pcfa_conv3x3_run"]
    core._WORK_TABLE["pcfa_conv3x3_run"] = lambda a: calls.append((a[6], a[7], a[8], a[9], a[10], a[11], bool(a[3]), bool(a[4])))
    core.set_work_recorder({})
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        st.optimizer.zero_grad()
        st._closure_body()
        torch.cuda.synchronize()
    core.set_work_recorder(None)
    core._WORK_TABLE["pcfa_conv3x3_run"] = orig
    evs = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA and
                  ("conv3x3_winograd_kernel" in e.name or "conv3x3_f43_kernel" in e.name or "f43_finish_kernel" in e.name)),
                 key=lambda e: e.time_range.start)
    evs = [e for e in evs]
    i = 0
    rows = collections.OrderedDict()
    for c in calls:
        if i >= len(evs):
            break
        e = evs[i]
        us, kind = e.time_range.elapsed_us(), ("f43" if "f43" in e.name else "f23")
        i += 1
        if "finish" not in e.name and i < len(evs) and "f43_finish" in evs[i].name:   # partial outputs + finish pass
            us += evs[i].time_range.elapsed_us()
            kind += "+split"
            i += 1
        r = rows.setdefault(c + (kind,), [0, 0.0])
        r[0] += 1
        r[1] += us
    tot = sum(v[1] for v in rows.values())
    print("%s %dx%d: %d conv3x3 calls, %.1f us of device time (matched %d of %d kernel events)" % (net, h, w, len(calls), tot, i, len(evs)))
    print("   B    K    N    H    W act mask add  kind       calls   mean us  total us  direct TF/s")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        B, K, N, H, W, act, mask, add, kind = k
        fl = 2.0 * 9 * K * N * B * H * W
        print("%4d %4d %4d %4d %4d %3d %4d %3d  %-10s %5d %9.1f %9.1f %9.1f" % (B, K, N, H, W, act, mask, add, kind, v[0], v[1] / v[0],
                                                                           v[1], fl / (v[1] / v[0]) / 1e6))


if __name__ == "__main__":
    main()
