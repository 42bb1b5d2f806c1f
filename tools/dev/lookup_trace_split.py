#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: duration of the lookup forward kernel inside the hipGraph replays (timed
region) versus inside the final eagerly launched measurement step (the launches that carry dispatch-attached events)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "corr_lookup_fwd_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n_eager = 132
eager, before = d[-n_eager:], d[:-n_eager]
print("lookup fwd launches: %d" % len(d))
print("final eager step (%d launches, events attached): mean %.2f us  median %.2f" %
      (len(eager), sum(eager) / len(eager), sorted(eager)[len(eager) // 2]))
g = before[-396:]
if g:
    print("graph replays before it (%d launches): mean %.2f us  median %.2f" % (len(g), sum(g) / len(g), sorted(g)[len(g) // 2]))
else:  # r02: inside the replays the lookup runs fused with convc1 (corr_lookup_convc1_fwd_kernel)
    allrows = [r for r in csv.DictReader(open(sys.argv[1])) if "corr_lookup_convc1_fwd_kernel" in r["Kernel_Name"]]
    allrows.sort(key=lambda r: int(r["Start_Timestamp"]))
    f = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in allrows]
    print("fused lookup+convc1 forward launches: %d, last 396 (graph replays): mean %.2f us  median %.2f" %
          (len(f), sum(f[-396:]) / len(f[-396:]), sorted(f[-396:])[len(f[-396:]) // 2]))
first = d[:132]
print("first eager warm-up step (%d launches, no events): mean %.2f us  median %.2f" %
      (len(first), sum(first) / len(first), sorted(first)[len(first) // 2]))
