#!/bin/bash
# round 4, GPU call 24: loss row kernel with its operands requested one row ahead — op tests, A/B (DL3_XENT_PREF=0 = the old form),
# then the driver's sequence on this tree (full -m gpu suite, smoke, default bench)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4x; mkdir -p $out
cd $REPO
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "xent or upsample" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -1 $out/pytest_ops.log; grep -h "^E  \|^FAILED" $out/pytest_ops.log | cut -c1-200 | head -8
bash tools/r4/ab.sh r4x/ab128 "--steps 12 --warmup 3" "1_old|DL3_XENT_PREF=0" "2_pref|DL3_DY_MAT=1" "3_old_again|DL3_XENT_PREF=0" | tee $out/ab128.txt
python - <<'PY'
import json
for v in ("1_old", "2_pref", "3_old_again"):
    r = [x for x in json.load(open("gpurun_out/r4x/ab128/%s.plan.json" % v))["rows"] if x["op"] == "dl3_upsample_softmax_xent_fold"]
    print(v, "xent_fold ms", [round(x["ms"], 3) for x in r])
PY
bash tools/r4/final_check.sh
