#!/bin/bash
# round 4, GPU call 4: fused backward kernel v2 (raw x staging, LDS epilogue operands) — op tests, A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4d; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fused" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -3 $out/pytest_ops.log
grep -h "^E  " $out/pytest_ops.log | head -10
timeout 1200 python -m pytest tests/test_gpu_model.py -q -k "mnv2_train_step or poison or adam" > $out/pytest_model.log 2>&1; echo "model rc $?"; tail -3 $out/pytest_model.log
grep -h "^E  " $out/pytest_model.log | cut -c1-300 | head -10
bash tools/r4/ab.sh r4d/ab128 "--steps 15 --warmup 3" \
  "1_fused0|DL3_FUSED_BWD=0" "2_fused1|DL3_FUSED_BWD=1" "3_fused1_wgs4096|DL3_FUSED_WGS=4096" "4_fused1_wgs1024|DL3_FUSED_WGS=1024" \
  "5_fused0_again|DL3_FUSED_BWD=0" "6_fused1_again|DL3_FUSED_BWD=1" | tee $out/ab128.txt
bash tools/r4/ab.sh r4d/ab16 "--steps 40 --warmup 3 --batch 16" "1_fused0|DL3_FUSED_BWD=0" "2_fused1|DL3_FUSED_BWD=1" | tee $out/ab16.txt
python - <<'PY'
import json
for f in ("2_fused1", "3_fused1_wgs4096"):
    rows = json.load(open("gpurun_out/r4d/ab128/%s.plan.json" % f))["rows"]
    for r in rows:
        if r["shape"].startswith("bwd-fused"):
            print(f, "%-36s %.3f ms %.2f TB/s" % (r["shape"], r["ms"], r["bytes"] / r["ms"] / 1e9))
PY
