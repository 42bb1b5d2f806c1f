#!/bin/bash
# On the GPU box: kernel mix of one captured closure of NET (rocprofv3 --kernel-trace over 50 graph replays)
# -> gpurun_out/<tag>_closure_profile_<net>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
NET=${2:-FlowNet2}
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_closure_$NET -- python3 $R/tools/closure_profile.py run $NET > $R/gpurun_out/${TAG}_closure_run_$NET.log 2>&1
rc=$?
cd $R
t=$(find gpurun_out/prof_closure_$NET -name "*kernel_trace.csv" | head -1)
python tools/closure_profile.py report $t 45 > gpurun_out/${TAG}_closure_profile_$NET.txt
rm -f $t
head -30 gpurun_out/${TAG}_closure_profile_$NET.txt | cut -c1-150
exit $rc
