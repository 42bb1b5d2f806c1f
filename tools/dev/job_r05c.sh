for i in 1 2 3; do python -m pytest tests/test_gpu_parity.py -q -x -k "gru_step or sepconv5_vs or gru_gate" 2>&1 | tail -2; done
python tools/bench_gru_step.py 2>&1 | grep -E "winograd F|sc5_wino" | cut -c1-100
