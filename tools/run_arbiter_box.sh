#!/bin/bash
# On the GPU box: fp64-arbiter records (tools/parity_arbiter.py) for a list of "NET:SEED:AT_STEPS[:VARIANTS]" items.
#   tools/run_arbiter_box.sh OUTDIR RAFT:6:0:default,f23 PWCNet:0:5 ...
OUT=$1; shift
mkdir -p $OUT
for item in "$@"; do
  IFS=: read NET SEED STEPS VARS <<< "$item"
  python tools/parity_arbiter.py run --net $NET --seed $SEED --at-steps $STEPS --gpu-variants ${VARS:-default} \
      --out $OUT 2>> $OUT/log_${NET}_${SEED}.txt
  echo "$item rc=$?"
done
