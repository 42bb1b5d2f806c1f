timeout -k 10 120 python tools/bench_lookup_conv.py 2>&1 | grep -E "fwd_kernel|bwd_kernel" | head -4
PCFA_HIP_LIB=$PWD/pcfa_amd/lib/libpcfa_hip_stamps.so timeout -k 10 120 python tools/dev/lc_stamps.py 2>&1 | grep median
