mkdir -p gpurun_out/r05_matrix
python tools/parity_matrix.py gpu --net PWCNet --seeds 0,1,2,3,4,5,6,7 --steps 20 --out gpurun_out/r05_matrix 2> gpurun_out/r05_matrix/gpu_PWC20.log; tail -1 gpurun_out/r05_matrix/gpu_PWC20.log
python tools/parity_matrix.py gpu --net PWCNet --seeds 0,1 --steps 50 --out gpurun_out/r05_matrix 2> gpurun_out/r05_matrix/gpu_PWC50.log; tail -1 gpurun_out/r05_matrix/gpu_PWC50.log
tools/run_arbiter_box.sh gpurun_out/r05_arbiter GMA:4:0
