#!/bin/bash
# On the GPU box: one bench line per network of BASELINE.json's configs (N = 1), appended to
# gpurun_out/<tag>_bench_nets.jsonl.  RAFT is the headline line bench.py prints by default.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd $R
: > gpurun_out/${TAG}_bench_nets.jsonl
for spec in "GMA 436x1024" "PWCNet 375x1242" "SpyNet 436x1024" "FlowNet2 436x1024"; do
  set -- $spec
  timeout -k 10 400 python bench.py --net $1 --size $2 --steps 3 --warmup 1 --cpu-closures 4 >> gpurun_out/${TAG}_bench_nets.jsonl 2>> gpurun_out/${TAG}_bench_nets.err || exit 1
done
python - <<PY
import json
for line in open("gpurun_out/${TAG}_bench_nets.jsonl"):
    d = json.loads(line)
    c = d.get("cpu_baseline") or {}
    print("%-70s %7.3f steps/s  %8.1f ms/step  closures/s %.1f  cpu port %.4f steps/s on %s threads" % (d["config"]["workload"][:70], d["value"], d["ms_per_step"], d["closure_evals_per_sec"], c.get("value", float("nan")), c.get("cores", "?")))
PY
