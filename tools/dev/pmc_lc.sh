#!/bin/bash
# Instruction-fetch / issue counters of the fused lookup kernels (tools/bench_lookup_conv.py), one --pmc pass per group.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc_counters.txt 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lc_$i -- \
      python3 $R/tools/bench_lookup_conv.py > $R/gpurun_out/pmc_lc_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_lc_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "convc1_fwd" in k or "convc1_bwd" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
