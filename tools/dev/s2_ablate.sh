#!/bin/bash
# Phase ablation of conv_s2_fwd_kernel (timing-only builds: tools/dev/build_variant.sh s2d<N> conv_strided.hip -DPCFA_S2_DBG=<N>)
for d in "" ${S2_ABLATE:-1 2 4 6 14}; do
  lib=pcfa_amd/lib/libpcfa_hip.so; [ -n "$d" ] && lib=pcfa_amd/lib/libpcfa_hip_s2d$d.so
  echo "== DBG=${d:-0}"
  PCFA_HIP_LIB=$PWD/$lib timeout -k 10 100 python tools/dev/bench_conv_s2.py --hip-only 2>&1 | grep -E "conv_s2_fwd" 
done
