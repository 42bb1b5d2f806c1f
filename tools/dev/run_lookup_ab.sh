#!/bin/bash
# On the GPU box: A/B the lookup-forward variants of tools/dev/liblookup_dev.so (stamps + rocprofv3 kernel times).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
V=${1:-0,1,2,3}
cd $R
timeout -k 10 200 python tools/dev/lookup_stamps.py stamps $V > gpurun_out/ab_stamps.txt 2>&1 || { tail -20 gpurun_out/ab_stamps.txt; exit 1; }
for mode in warm cold; do
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab_$mode -- python3 $R/tools/dev/lookup_stamps.py time $mode $V > $R/gpurun_out/ab_time_$mode.log 2>&1
  cd $R
  f=$(find gpurun_out/prof_ab_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; grep "variant" gpurun_out/ab_time_$mode.log
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "corr_lookup" in r["Name"]:
        print("%-70s calls %4s avg %7.2f us min %7.2f max %7.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  find gpurun_out/prof_ab_$mode -name "*kernel_trace.csv" -delete
done
