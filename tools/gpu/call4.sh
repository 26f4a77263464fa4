mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q -rf -k "not fullsize and not parallel" > gpurun_out/r2d/pytest_fast.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest_fast.log
tail -4 gpurun_out/r2d/pytest_fast.log
run() { # name, env..., batch
  local name=$1; shift; local B=$1; shift
  env "$@" python bench.py --batch $B --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > gpurun_out/r2d/$name.json 2> gpurun_out/r2d/$name.err
  echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2d/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r['config']['launches_per_step'])" 2>&1 | tail -1)"
}
for B in 2 16 64; do
  run b${B}_tails0 $B DL3_TAILS=0
  run b${B}_maxp16 $B DL3_TAIL_MAXP=16
  run b${B}_maxp48 $B DL3_TAIL_MAXP=48
  run b${B}_maxp128 $B DL3_TAIL_MAXP=128
  run b${B}_maxp100000 $B DL3_TAIL_MAXP=100000
done
run b64_prio 64 DL3_TAILS=0 DL3_GEMM_TUNE=1
run b64_py1024 64 DL3_TAILS=0 DL3_GEMM_PY=1024
run b64_py4096 64 DL3_TAILS=0 DL3_GEMM_PY=4096
run b64_py512 64 DL3_TAILS=0 DL3_GEMM_PY=512
run b16_prio 16 DL3_TAILS=0 DL3_GEMM_TUNE=1
