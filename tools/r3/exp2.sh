#!/bin/bash
out=gpurun_out/r3b; mkdir -p $out
python tools/r3/bn_probe.py 256 xception > $out/bn_probe_x256.log 2>&1
python tools/r3/bn_probe.py 256 mobilenetv2 > $out/bn_probe_m256.log 2>&1
tail -20 $out/bn_probe_x256.log; tail -6 $out/bn_probe_m256.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" > $out/b_$name.json 2> $out/b_$name.err; python - $out/b_$name.json <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2), r["config"].get("backward_fork"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
}
for b in 2 16; do
run nofork_b$b DL3_FORK=0 $B --batch $b
run fork_b$b DL3_FORK=1 $B --batch $b
run fork_w512_b$b DL3_FORK=1 DL3_WGRAD_WGS=512 $B --batch $b
done
run nofork_b192 DL3_FORK=0 $B --batch 192
run nofork_b256 DL3_FORK=0 $B --batch 256
