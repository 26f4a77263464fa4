#!/bin/bash
out=gpurun_out/r3o; mkdir -p $out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for sg in 0 25 50 100 150; do run sg$sg DL3_GEMM_STAGGER=$sg $B; done
run sg0b DL3_GEMM_STAGGER=0 $B
