mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -q -rf -k "not fullsize and not parallel" > gpurun_out/r2c/pytest_fast.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pytest_fast.log
tail -12 gpurun_out/r2c/pytest_fast.log
for B in 64 16 2; do
  for T in 1 0; do
    DL3_TAILS=$T python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2c/bench_b${B}_t${T}.json 2> gpurun_out/r2c/bench_b${B}_t${T}.err
    echo "B=$B tails=$T: $(python -c "import json;r=json.load(open('gpurun_out/r2c/bench_b${B}_t${T}.json'));print(round(r['value'],1), r['ms_per_step'], r['config']['launches_per_step'])" 2>&1 | tail -1)"
  done
done
for O in 1 2 3; do
  DL3_GEMM_OCC1=$O python bench.py --batch 64 --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2c/bench_b64_occ${O}.json 2> gpurun_out/r2c/bench_b64_occ${O}.err
  echo "B=64 occ1=$O: $(python -c "import json;r=json.load(open('gpurun_out/r2c/bench_b64_occ${O}.json'));print(round(r['value'],1), r['ms_per_step'])" 2>&1 | tail -1)"
done
DL3_GEMM_OCC1=3 python bench.py --batch 64 --no-cpu-baseline --steps 10 --warmup 3 --plan-json gpurun_out/r2c/plan_b64_occ3.json > gpurun_out/r2c/bench_b64_occ3_plan.json 2>/dev/null
python bench.py --batch 64 --no-cpu-baseline --steps 10 --warmup 3 --plan-json gpurun_out/r2c/plan_b64_occ0.json > gpurun_out/r2c/bench_b64_occ0_plan.json 2>/dev/null
DL3_GEMM_OCC1=3 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "pwconv" 2>&1 | tail -3
