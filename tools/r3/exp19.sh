#!/bin/bash
out=gpurun_out/r3s; mkdir -p $out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
run base1 $B
run prio1 DL3_LIBPATH=$PWD/build_variants/libdl3_prio1.so $B
run prio3 DL3_LIBPATH=$PWD/build_variants/libdl3_prio3.so $B
run base2 $B
run prio3b DL3_LIBPATH=$PWD/build_variants/libdl3_prio3.so $B
run base_b16 $B --batch 16
run prio3_b16 DL3_LIBPATH=$PWD/build_variants/libdl3_prio3.so $B --batch 16
