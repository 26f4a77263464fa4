#!/bin/bash
out=gpurun_out/r3k; mkdir -p $out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline --backbone xception --os 8"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],2), round(r["ms_per_step"],2))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
run x_pad736_b16 DL3_CHANNEL_PAD=32 $B --batch 16
run x_pad768_b16 DL3_CHANNEL_PAD=1 $B --batch 16
run x_pad736_b32 DL3_CHANNEL_PAD=32 $B --batch 32
run x_pad768_b32 DL3_CHANNEL_PAD=1 $B --batch 32
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -x -m gpu -k "xception" 2>&1 | tail -4
