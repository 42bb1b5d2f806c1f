#!/bin/bash
# PWC-Net's conv3x3 calls (tools/dev/conv_calls_timed.py) under the K-slice settings of the F(2x2,3x3) kernel:
# policy as shipped, slices off, and F(2x2,3x3) forced everywhere with 0 / auto / 2 / 4 / 8 slices.
# usage (GPU box): tools/dev/conv_ksl_ab.sh OUT.txt
out=${1:-gpurun_out/conv_ksl_ab.txt}
: > "$out"
run() { echo "=== $*" >> "$out"; env "$@" python tools/dev/conv_calls_timed.py PWCNet 375x1242 joint 2>&1 | grep -v "^-->" >> "$out" || exit 1; }
run PCFA_X=policy
run PCFA_CONV3X3_KSL=0
run PCFA_CONV3X3_ALGO=f23 PCFA_CONV3X3_KSL=0
run PCFA_CONV3X3_ALGO=f23
run PCFA_CONV3X3_ALGO=f23 PCFA_CONV3X3_KSL=2
run PCFA_CONV3X3_ALGO=f23 PCFA_CONV3X3_KSL=4
run PCFA_CONV3X3_ALGO=f23 PCFA_CONV3X3_KSL=8
