#!/bin/bash
# On the GPU box: kernel_grids.sh NET HxW SUBSTRING -> per (kernel, grid) device time of 5 eager closures (3 counted)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NET=${1:-RAFT}; SIZE=${2:-436x1024}; SUB=${3:-conv3x3}
mkdir -p $R/gpurun_out/kg
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kg/trace_$NET -- python3 $R/tools/dev/closure_eager.py $NET $SIZE 5 > /dev/null 2> $R/gpurun_out/kg/err_$NET.txt
cd $R
t=$(find gpurun_out/kg/trace_$NET -name "*kernel_trace.csv" | head -1)
python tools/dev/kernel_grids.py $t "$SUB" 5 > gpurun_out/kg/grids_${NET}.txt
rm -rf gpurun_out/kg/trace_$NET
cat gpurun_out/kg/grids_${NET}.txt
