#!/bin/bash
# On the GPU box: run tools/bench_kernels.py under rocprofv3 and print the per-kernel device times of
# the pcfa_amd kernels only (host-side event timings of sub-20us kernels are launch-bound and useless).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_kb_$1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_kernels.py > $R/gpurun_out/kb_$1.log 2>&1
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py $f 0 | grep -v "^#" | cut -c1-150
find $OUT -name "*kernel_trace.csv" -delete
