#!/bin/bash
out=gpurun_out/r3n; mkdir -p $out
SP=$PWD/build_variants/libdl3_schedpipe.so
DL3_LIBPATH=$SP timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pwconv or gemm or pw_" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
run base1 $B
run sp1 DL3_LIBPATH=$SP $B
run base2 $B
run sp2 DL3_LIBPATH=$SP $B
run base_b16 $B --batch 16
run sp_b16 DL3_LIBPATH=$SP $B --batch 16
run base_x $B --backbone xception --os 8 --batch 16 --steps 10
run sp_x DL3_LIBPATH=$SP $B --backbone xception --os 8 --batch 16 --steps 10
