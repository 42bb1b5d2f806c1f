#!/bin/bash
# HBM traffic of every hand-written kernel of one eager RAFT attack step at 436x1024 (graph off: one dispatch row per launch),
# the Winograd kernels per launch shape: two rocprofv3 --pmc passes.  bash tools/pmc_traffic_raft.sh <tag>
export TMPDIR=/tmp
export PCFA_BENCH_NO_TRACER=1
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_raft}
mkdir -p $R/gpurun_out
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-pwcnet-leg --no-gma-leg \
      --no-pairs-in-flight-leg --no-shared-forward-leg > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE > gpurun_out/pmc_${TAG}_traffic.json
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_${TAG}_traffic.json'))
for k,v in sorted(d.items(), key=lambda kv: -kv[1]['traffic_bytes']*kv[1]['launches'])[:40]:
    print('%-90s %5d  fetch %8.2f MB  write %8.2f MB' % (k[:90], v['launches'], (v['fetch_bytes'] or 0)/1e6, (v['write_bytes'] or 0)/1e6))
"
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +8M -delete
