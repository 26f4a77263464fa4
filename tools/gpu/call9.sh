mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "a_stationary or pwconv" > gpurun_out/r2i/pytest_astat.log 2>&1; tail -12 gpurun_out/r2i/pytest_astat.log
run() { local name=$1; shift; local B=$1; shift
  env "$@" python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2i/$name.json 2> gpurun_out/r2i/$name.err
  echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2i/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r['config']['final_loss'])" 2>&1 | tail -1)"; }
run b64_astat0 64 DL3_GEMM_ASTAT=0
for K in 16 32 64 96 160; do run b64_astat$K 64 DL3_GEMM_ASTAT=$K; done
run b128_astat0 128 DL3_GEMM_ASTAT=0
run b128_astat64 128 DL3_GEMM_ASTAT=64
DL3_GEMM_ASTAT=64 python bench.py --batch 64 --no-cpu-baseline --steps 10 --warmup 3 --plan-json gpurun_out/r2i/plan_astat64.json > gpurun_out/r2i/bench_astat64.json 2>/dev/null
