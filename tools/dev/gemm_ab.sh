#!/bin/bash
# A/B of the fp32 MFMA GEMM core (csrc/corr_pyramid.hip) at the shapes of the path: K-chunk 16 / 32, the next stage's LDS
# write after the last MFMA or in the middle of the MFMA block (tools/bench_gemm.py per variant)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/r04_gemm_ab.txt
: > $OUT
echo "=== product build" >> $OUT
python tools/bench_gemm.py 2>/dev/null | grep TFLOP >> $OUT
for v in "bk32 -DPCFA_GEMM_BK=32" "mid4 -DPCFA_GEMM_STORE_MID=1 -DPCFA_GEMM_STORE_MID_AT=4" "mid8 -DPCFA_GEMM_STORE_MID=1 -DPCFA_GEMM_STORE_MID_AT=8" "bk32mid8 -DPCFA_GEMM_BK=32 -DPCFA_GEMM_STORE_MID=1 -DPCFA_GEMM_STORE_MID_AT=8" "bk32mid16 -DPCFA_GEMM_BK=32 -DPCFA_GEMM_STORE_MID=1 -DPCFA_GEMM_STORE_MID_AT=16"; do
  set -- $v; n=$1; shift
  tools/dev/build_variant.sh gemm_$n corr_pyramid.hip "$@" > /dev/null 2>&1 || { echo "build $n failed" >> $OUT; continue; }
  echo "=== $n ($*)" >> $OUT
  PCFA_HIP_LIB=$R/pcfa_amd/lib/libpcfa_hip_gemm_$n.so python tools/bench_gemm.py 2>/dev/null | grep TFLOP >> $OUT
done
cat $OUT
