#!/bin/bash
# A/B of where sc5_wino_kernel writes the next chunk's patch to LDS (PCFA_SC5W_STORE_AT = 0..4; 4 = after the last MFMA)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/r04_sc5w_store_ab.txt
: > $OUT
for N in 4 0 1 2; do
  tools/dev/build_variant.sh sc5ws$N sepconv5_wino.hip -DPCFA_SC5W_STORE_AT=$N > /dev/null 2>&1 || { echo "build $N failed" >> $OUT; continue; }
  echo "=== PCFA_SC5W_STORE_AT=$N" >> $OUT
  PCFA_HIP_LIB=$R/pcfa_amd/lib/libpcfa_hip_sc5ws$N.so python tools/bench_gru_step.py 2>/dev/null | grep -E "winograd F|sc5_wino|rel L2" >> $OUT
done
cut -c1-120 $OUT
