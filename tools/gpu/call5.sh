mkdir -p gpurun_out/r2e
run() { # name, batch, env...
  local name=$1; shift; local B=$1; shift
  env "$@" python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2e/$name.json 2> gpurun_out/r2e/$name.err
  echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2e/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r['config']['launches_per_step'])" 2>&1 | tail -1)"
}
run b64_base 64 X=1
for PY in 256 384 512 640 768; do run b64_py$PY 64 DL3_GEMM_PY=$PY; done
for W in 256 512 2048 4096; do run b64_wg$W 64 DL3_WGRAD_WGS=$W; done
for D in 512 1024 4096 8192; do run b64_dw$D 64 DL3_DW_BLOCKS=$D; done
run b64_py512_wg512 64 DL3_GEMM_PY=512 DL3_WGRAD_WGS=512
run b16_base 16 X=1
run b16_py512 16 DL3_GEMM_PY=512
run b16_py1024 16 DL3_GEMM_PY=1024
run b2_base 2 X=1
run b2_py512 2 DL3_GEMM_PY=512
# where does a small-batch step go?
python bench.py --batch 2 --no-cpu-baseline --steps 20 --warmup 5 --plan-json gpurun_out/r2e/plan_b2.json > gpurun_out/r2e/bench_b2_plan.json 2>/dev/null
python bench.py --batch 16 --no-cpu-baseline --steps 20 --warmup 5 --plan-json gpurun_out/r2e/plan_b16.json > gpurun_out/r2e/bench_b16_plan.json 2>/dev/null
# cfg4: where does the Xception step go?
python bench.py --backbone xception --os 8 --batch 8 --no-cpu-baseline --steps 5 --warmup 2 --plan-json gpurun_out/r2e/plan_xc8_b8.json > gpurun_out/r2e/bench_xc8_b8.json 2> gpurun_out/r2e/bench_xc8_b8.err
tail -c 600 gpurun_out/r2e/bench_xc8_b8.json
