#!/bin/bash
out=gpurun_out/r3v; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
V=$PWD/build_variants
for rep in 1 2; do
run base_$rep DL3_LIBPATH=$V/libdl3_base.so $B
run new_$rep DL3_LIBPATH=$V/libdl3_new.so $B
run directmm_$rep DL3_LIBPATH=$V/libdl3_direct_minmax.so $B
done
run base_b16 DL3_LIBPATH=$V/libdl3_base.so $B --batch 16
run new_b16 DL3_LIBPATH=$V/libdl3_new.so $B --batch 16
run base_x DL3_LIBPATH=$V/libdl3_base.so $B --backbone xception --os 8 --batch 16 --steps 10
run new_x DL3_LIBPATH=$V/libdl3_new.so $B --backbone xception --os 8 --batch 16 --steps 10
