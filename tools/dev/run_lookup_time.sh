#!/bin/bash
# On the GPU box: rocprofv3 durations of the product lookup kernels for the library in $PCFA_HIP_LIB (warm / cold).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in warm cold; do
  cd /tmp
  rm -rf $R/gpurun_out/prof_lookup_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lookup_$mode -- python3 $R/tools/dev/lookup_stamps.py time $mode > $R/gpurun_out/lookup_time_$mode.log 2>&1
  cd $R
  f=$(find gpurun_out/prof_lookup_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "corr_lookup" in r["Name"]:
        print("%-70s calls %4s avg %7.2f us min %7.2f max %7.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  find gpurun_out/prof_lookup_$mode -name "*kernel_trace.csv" -delete
done
