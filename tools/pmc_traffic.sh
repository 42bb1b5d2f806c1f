#!/bin/bash
# HBM traffic of the pcfa_amd kernels from rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing domains besides
# --kernel-trace, program directly after `--`.  Run on the GPU box:  bash tools/pmc_traffic.sh <tag>
export TMPDIR=/tmp
export PCFA_BENCH_NO_TRACER=1  # rocprofv3 owns the tracer in these runs
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/pmc_${TAG}_traffic.json
cat gpurun_out/pmc_${TAG}_traffic.json
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +8M -delete
