#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the default bench command -> gpurun_out/<tag>_bench_kernel_stats.txt,
# plus the per-closure kernel mix of the graph replays.
export TMPDIR=/tmp
export PCFA_BENCH_NO_TRACER=1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
shift
cd /tmp
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof_bench.err
cd $R
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python tools/summarize_rocprof.py $f 45 > gpurun_out/${TAG}_bench_kernel_stats.txt
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python tools/closure_profile.py report $t 40 > gpurun_out/${TAG}_closure_profile.txt
rm -f $t
head -12 gpurun_out/${TAG}_closure_profile.txt | cut -c1-140
