#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for dbg in 0 1 2 4 8 9 11 15; do
  echo -n "bwd dbg=$dbg (1=no X loads 2=no compute 4=no stores 8=no taps): "
  PCFA_SC_DBG=$dbg timeout -k 5 120 python $R/tools/bench_scorr.py kitti 20 2>/dev/null | grep "device time" | awk "{print \$11}" | tr '\n' ' '
  echo
done
