mkdir -p gpurun_out/r2j
DL3_GEMM_ASTAT=16 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "a_stationary" 2>&1 | tail -2
run() { local name=$1; shift; local B=$1; shift
  env "$@" python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2j/$name.json 2> gpurun_out/r2j/$name.err
  echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2j/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r['config']['final_loss'])" 2>&1 | tail -1)"; }
run b64_off 64 DL3_GEMM_ASTAT=0
run b64_64_160 64 DL3_GEMM_ASTAT=64 DL3_GEMM_ASTAT_MAXK=160
run b64_64_96 64 DL3_GEMM_ASTAT=64 DL3_GEMM_ASTAT_MAXK=96
run b64_32_96 64 DL3_GEMM_ASTAT=32 DL3_GEMM_ASTAT_MAXK=96
run b64_16_96 64 DL3_GEMM_ASTAT=16 DL3_GEMM_ASTAT_MAXK=96
run b64_160_160 64 DL3_GEMM_ASTAT=160 DL3_GEMM_ASTAT_MAXK=160
run b16_off 16 DL3_GEMM_ASTAT=0
run b16_64_96 16 DL3_GEMM_ASTAT=64 DL3_GEMM_ASTAT_MAXK=96
run b128_off 128 DL3_GEMM_ASTAT=0
run b128_64_96 128 DL3_GEMM_ASTAT=64 DL3_GEMM_ASTAT_MAXK=96
DL3_GEMM_ASTAT=64 python bench.py --batch 64 --no-cpu-baseline --steps 10 --warmup 3 --plan-json gpurun_out/r2j/plan_astat64_160.json > /dev/null 2>&1
