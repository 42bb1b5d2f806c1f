mkdir -p gpurun_out/r05_matrix
python -m pytest tests/test_gpu_parity.py -q -k "pairs_in_flight or conv3x3_winograd_vs_oracle or conv3x3_f43" 2>&1 | tail -2
python tools/parity_matrix.py gpu --net RAFT --seeds 0,1,2,3,4,5,6,7 --steps 20 --out gpurun_out/r05_matrix 2> gpurun_out/r05_matrix/gpu_RAFT.log; tail -2 gpurun_out/r05_matrix/gpu_RAFT.log
python tools/parity_matrix.py gpu --net GMA --seeds 0,1 --steps 20 --out gpurun_out/r05_matrix 2> gpurun_out/r05_matrix/gpu_GMA01.log; tail -2 gpurun_out/r05_matrix/gpu_GMA01.log
tools/run_matrix_box.sh GMA "2,3" 20 16
