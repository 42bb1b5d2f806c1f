mkdir -p gpurun_out/r05w
python -m pytest tests -m gpu -q > gpurun_out/r05w/pytest_gpu.txt 2>&1; tail -2 gpurun_out/r05w/pytest_gpu.txt
bash tools/prof_bench.sh r05 --steps 5 --warmup 2 --no-pwcnet-leg --no-gma-leg --no-pairs-in-flight-leg --no-shared-forward-leg > gpurun_out/r05w/prof.txt 2>&1; head -12 gpurun_out/r05_bench_kernel_stats.txt | cut -c1-140
