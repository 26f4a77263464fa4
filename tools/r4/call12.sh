#!/bin/bash
# round 4, GPU call 12: fused both-gradient kernel with four workgroups per CU for the <= 4-block shapes (launch bounds 256, 4)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4l; mkdir -p $out
cd $REPO
V=$REPO/build_variants/libdl3_fused_b4.so
DL3_LIBPATH=$V timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fused" > $out/pytest_ops.log 2>&1; echo "ops(b4) rc $?"; tail -1 $out/pytest_ops.log
bash tools/r4/ab.sh r4l/ab128 "--steps 15 --warmup 3" "1_b3|DL3_DY_MAT=1" "2_b4|DL3_LIBPATH=$V" "3_b3_again|DL3_DY_MAT=1" "4_b4_again|DL3_LIBPATH=$V" | tee $out/ab128.txt
python - <<'PY'
import json
for f in ("1_b3", "2_b4", "3_b3_again", "4_b4_again"):
    rows = json.load(open("gpurun_out/r4l/ab128/%s.plan.json" % f))["rows"]
    print(f, " ".join("%s:%.3f" % (r["shape"].replace("bwd-fused ", "").replace(" ", ""), r["ms"]) for r in rows if r["shape"].startswith("bwd-fused")))
PY
