#!/bin/bash
# Phase ablation of sc5_wino_kernel (timing-only builds: tools/dev/build_variant.sh sc5w<N> sepconv5_wino.hip -DPCFA_SC5W_DBG=<N>):
# device time of one SepConvGRU update (tools/bench_gru_step.py) per variant -> gpurun_out/r04_sc5w_ablation.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/r04_sc5w_ablation.txt
: > $OUT
for N in 0 1 2 4 8 16 31; do
  tools/dev/build_variant.sh sc5w$N sepconv5_wino.hip -DPCFA_SC5W_DBG=$N > /dev/null 2>&1 || { echo "build $N failed" >> $OUT; continue; }
  echo "=== PCFA_SC5W_DBG=$N (1 no barrier, 2 no transform, 4 no LDS reads, 8 no weight loads, 16 no patch loads/stores)" >> $OUT
  PCFA_HIP_LIB=$R/pcfa_amd/lib/libpcfa_hip_sc5w$N.so python tools/bench_gru_step.py 2>/dev/null | grep -E "winograd F|sc5_wino" >> $OUT
done
cat $OUT | cut -c1-120
