mkdir -p gpurun_out/r05n
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05n/bench_line.json 2> gpurun_out/r05n/bench_err.txt; echo rc=$?; cp bench_detail.json gpurun_out/r05n/bench_detail.json
wc -c gpurun_out/r05n/bench_line.json; cat gpurun_out/r05n/bench_line.json
bash tools/prof_bench.sh r05 --steps 5 --warmup 2 --no-pwcnet-leg --no-gma-leg --no-pairs-in-flight-leg --no-shared-forward-leg > gpurun_out/r05n/prof.txt 2>&1; tail -14 gpurun_out/r05n/prof.txt | cut -c1-150
