#!/bin/bash
out=gpurun_out/r3e; mkdir -p $out
DL3_GEMM_PF=2 timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pwconv or gemm or pw_" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for pf in 0 1 2 3 4 6; do run pf$pf DL3_GEMM_PF=$pf $B; done
run pf0_again DL3_GEMM_PF=0 $B
for pf in 0 2 4; do run x_pf$pf DL3_GEMM_PF=$pf $B --backbone xception --os 8 --batch 16; done
