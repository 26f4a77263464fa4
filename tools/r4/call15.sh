#!/bin/bash
# round 4, GPU call 15: what do the BatchNorm finalize launches cost inside the replayed graph?  (timing probe: launches dropped)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4o; mkdir -p $out
cd $REPO
S="dl3_bn_finalize,dl3_bn_bwd_finalize"
run() {  # name, skip list, bench args
  DL3_PROBE_SKIP=$2 timeout 600 python tools/r4/skip_ops_probe.py $3 --no-cpu-baseline --no-split-leg --no-legs --no-roofline > $out/$1.json 2> $out/$1.err
  python - "$out/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-22s %9.1f img/s %9.3f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run b2_base "" "--batch 2 --steps 100 --warmup 3"
run b2_nofin "$S" "--batch 2 --steps 100 --warmup 3"
run b2_base_again "" "--batch 2 --steps 100 --warmup 3"
run b2_nofin_again "$S" "--batch 2 --steps 100 --warmup 3"
run b16_base "" "--batch 16 --steps 40 --warmup 3"
run b16_nofin "$S" "--batch 16 --steps 40 --warmup 3"
run b128_base "" "--steps 12 --warmup 3"
run b128_nofin "$S" "--steps 12 --warmup 3"
