mkdir -p gpurun_out/r05b
python tools/bench_gru_step.py > gpurun_out/r05b/gru_wlds.txt 2>&1 &&
PCFA_HIP_LIB=$PWD/pcfa_amd/lib/libpcfa_hip_sc5w_regs.so python tools/bench_gru_step.py > gpurun_out/r05b/gru_regs.txt 2>&1 &&
grep -E "winograd F|sc5_wino" gpurun_out/r05b/gru_wlds.txt | cut -c1-110 && echo ---- && grep -E "winograd F|sc5_wino" gpurun_out/r05b/gru_regs.txt | cut -c1-110 &&
python -m pytest tests/test_gpu_parity.py -x -q -k "sepconv5 or gru_step or gru_gate or pairs_in_flight or non_finite" > gpurun_out/r05b/pytest.txt 2>&1; tail -3 gpurun_out/r05b/pytest.txt
python tools/dev/conv_shapes.py RAFT 436x1024 > gpurun_out/r05b/conv_shapes_raft.txt 2>&1; tail -30 gpurun_out/r05b/conv_shapes_raft.txt
