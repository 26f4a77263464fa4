mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q -rf -x -k "not fullsize" > gpurun_out/r2b/pytest_fast.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_fast.log
tail -3 gpurun_out/r2b/pytest_fast.log
for B in 64 16 2; do
  for T in 1 0; do
    DL3_TAILS=$T python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2b/bench_b${B}_t${T}.json 2> gpurun_out/r2b/bench_b${B}_t${T}.err
    echo "B=$B tails=$T: $(python -c "import json;r=json.load(open('gpurun_out/r2b/bench_b${B}_t${T}.json'));print(round(r['value'],1), r['ms_per_step'], r['config']['launches_per_step'])" 2>&1 | tail -1)"
  done
done
python -m pytest tests/test_gpu_fullsize.py -q -rf --durations=8 > gpurun_out/r2b/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pytest_full.log
tail -15 gpurun_out/r2b/pytest_full.log
