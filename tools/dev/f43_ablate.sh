#!/bin/bash
# Phase ablation of conv3x3_f43_kernel (timing-only builds, tools/dev/build_variant.sh f43d<N> conv3x3_f43.hip -DPCFA_F43_DBG=<N>)
for d in "" 1 8 11; do
  lib=pcfa_amd/lib/libpcfa_hip.so; [ -n "$d" ] && lib=pcfa_amd/lib/libpcfa_hip_f43d$d.so
  echo "== DBG=${d:-0}"
  PCFA_HIP_LIB=$PWD/$lib PCFA_CONV3X3_ALGO=f43 timeout -k 10 100 python tools/dev/bench_conv3x3.py --no-lib -v --few 2>&1 | grep -E "f43_kernel"
done
