mkdir -p gpurun_out/r05h
python -m pytest tests/test_gpu_parity.py -q -k "scorr or spatial_corr or cost_volume or dropin or pwc or PWC or pairs_in_flight or conv3x3_winograd_vs_oracle" 2>&1 | tail -2
python tools/bench_scorr.py kitti 200 > gpurun_out/r05h/bench_scorr.txt 2>&1; tail -14 gpurun_out/r05h/bench_scorr.txt
bash tools/pmc_traffic_pwc.sh r05_pwc_xcd > gpurun_out/r05h/pmc.txt 2>&1
tools/run_matrix_box.sh GMA "4,5" 20 16
