#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for bth in 2 4; do
echo "bwd tile rows $bth:"; PCFA_SC_BTH=$bth timeout -k 5 120 python $R/tools/bench_scorr.py kitti 20 2>/dev/null | grep "device time\|sum of 5 levels, DEV"
done
echo "default:"; timeout -k 5 120 python $R/tools/bench_scorr.py kitti 20 2>/dev/null | grep "device time\|sum of 5 levels, DEV"
for cfg in "0 2 1000" "2 8 1000"; do set -- $cfg
for dbg in 0 1 2 4 7; do
  echo -n "tile=$1 ns=$2 cr=$3 dbg=$dbg (1=no loads 2=no compute 4=no stores): "
  PCFA_SC_DBG=$dbg PCFA_SC_TILE=$1 PCFA_SC_NS=$2 PCFA_SC_CR=$3 timeout -k 5 120 python $R/tools/bench_scorr.py kitti 20 2>/dev/null | grep "device time" | awk '{print $6}' | tr '\n' ' '
  echo
done; done
