#!/usr/bin/env python3
"""Kernel mix of ONE closure evaluation (forward + loss + backward) at the bench shape.

  closure_profile.py run [NET]        replays the captured closure 50 times (run under rocprofv3 --kernel-trace)
  closure_profile.py report <kernel_trace.csv> [top]   per-kernel device time of the LAST 40 replays, per closure
"""
import csv
import os
import sys


def run(net):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    st = bench.AttackStepper(net, 436, 1024, torch.device("cuda", 0), seed=0)
    st.step()
    st.enable_graph()
    torch.cuda.synchronize()
    for _ in range(50):
        st.graphed()
    torch.cuda.synchronize()


def report(path, top):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the 50 replays are the tail of the trace: find the period by the loss kernel, which runs once per closure
    marks = [i for i, r in enumerate(rows) if "loss_final_kernel" in r["Kernel_Name"]]
    marks = marks[-41:]
    sel = rows[marks[0] + 1: marks[-1] + 1]
    n = len(marks) - 1
    agg = {}
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0])
        a[0] += 1
        a[1] += d
    tot = sum(a[1] for a in agg.values())
    wall = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    print("# %d closures: %.3f ms device time per closure, %.3f ms wall per closure, %d launches per closure"
          % (n, tot / n / 1e6, wall / n / 1e6, len(sel) // n))
    print("%-100s %8s %10s %9s %7s" % ("kernel", "calls", "us/closure", "avg_us", "pct"))
    for name, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-100s %8.1f %10.1f %9.2f %7.2f" % (name[:100], c / n, d / n / 1e3, d / c / 1e3, 100.0 * d / tot))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "RAFT")
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
