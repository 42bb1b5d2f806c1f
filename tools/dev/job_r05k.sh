tools/run_arbiter_box.sh gpurun_out/r05_arbiter PWCNet:6:5 PWCNet:7:1 GMA:0:4
python tools/bench_conv3x3.py --no-lib 2>/dev/null | cut -c1-200 > gpurun_out/r05_arbiter/../r05k_conv3x3.txt
PCFA_HIP_LIB=$PWD/pcfa_amd/lib/libpcfa_hip_c3noraw.so python tools/bench_conv3x3.py --no-lib 2>/dev/null | cut -c1-200 > gpurun_out/r05k_conv3x3_noraw.txt
python tools/closure_timeline.py RAFT 436x1024 > gpurun_out/r05k_timeline.txt 2>/dev/null; tail -1 gpurun_out/r05k_timeline.txt
