#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid, workgroup): kernel_grids.py TRACE.csv [substring] [closures]"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: [0, 0.0])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
ncl = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if sub not in name:
            continue
        wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1)) * int(r.get("Workgroup_Size_Z", 1))
        grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
        k = (name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70], grid // wg, wg)
        rows[k][0] += 1
        rows[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-70s %7s %5s %7s %9s %8s" % ("kernel", "wgs", "thr", "calls", "us/clos", "avg us"))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-70s %7d %5d %7.1f %9.1f %8.2f" % (k[0], k[1], k[2], v[0] / ncl, v[1] / ncl, v[1] / v[0]))
