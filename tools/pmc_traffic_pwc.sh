#!/bin/bash
# HBM traffic of PWC-Net's cost-volume / warp kernels per level: two rocprofv3 --pmc passes over one
# eager PWC-Net attack step at 375x1242 (graph off: one dispatch row per launch).  bash tools/pmc_traffic_pwc.sh <tag>
export TMPDIR=/tmp
export PCFA_BENCH_NO_TRACER=1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_pwc}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- \
      python3 $R/bench.py --net PWCNet --size 375x1242 --steps 1 --warmup 1 --no-cpu-baseline --no-graph \
      > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/pmc_${TAG}_traffic.json
cat gpurun_out/pmc_${TAG}_traffic.json
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +8M -delete
