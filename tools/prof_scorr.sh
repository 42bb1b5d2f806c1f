#!/bin/bash
# On the GPU box: tools/bench_scorr.py under rocprofv3 --kernel-trace --stats; prints the cost-volume kernels' device times.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_scorr_$1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bench_scorr.py ${2:-kitti} 200 > $R/gpurun_out/scorr_$1.log 2>&1
cd $R
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py $f 0 | grep -v "^#" | grep -i "scorr\|kernel  " | cut -c1-170
find $OUT -name "*kernel_trace.csv" -delete
