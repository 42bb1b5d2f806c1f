mkdir -p gpurun_out/r05d
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05d/bench_line.json 2> gpurun_out/r05d/bench_err.txt; echo rc=$? ; cp bench_detail.json gpurun_out/r05d/bench_detail.json
wc -c gpurun_out/r05d/bench_line.json; cat gpurun_out/r05d/bench_line.json
for v in f23 f43big; do
  PCFA_CONV3X3_ALGO=$v python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pwcnet-leg --no-gma-leg --no-pairs-in-flight-leg --no-shared-forward-leg > gpurun_out/r05d/bench_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/r05d/bench_$v.json'));print('$v', d['value'], d['ms_per_step'])"
done
bash tools/pmc_traffic_pwc.sh r05_pwc > gpurun_out/r05d/pmc_pwc.txt 2>&1; tail -5 gpurun_out/r05d/pmc_pwc.txt
