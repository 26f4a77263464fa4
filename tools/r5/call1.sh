#!/bin/bash
# round 5, GPU call 1: new kernels — op tests, per-shape microbenchmarks, whole-step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
O=gpurun_out/c1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary or fused or pwconv_fwd" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "fused or determinism or poisoned or mnv2_train_step" > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg2_cfg3 and deeplab" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
DL3_FWD_WS=0 timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_tiled.log 2>&1
timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_ws.log 2>&1
timeout 600 python tools/r5/pw_hbm_bench.py fused > $O/mb_fused.log 2>&1
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
DL3_FWD_WS=0 DL3_FUSED_V=1 timeout 300 python bench.py $B --plan-json $O/plan_base.json > $O/bench_base.json 2> $O/bench_base.err
timeout 300 python bench.py $B --plan-json $O/plan_new.json > $O/bench_new.json 2> $O/bench_new.err
DL3_FUSED_OCC=3 timeout 300 python bench.py $B --plan-json $O/plan_occ3.json > $O/bench_occ3.json 2> $O/bench_occ3.err
DL3_FWD_WS=0 timeout 300 python bench.py $B > $O/bench_fusedonly.json 2> $O/bench_fusedonly.err
cat $O/status.txt
tail -3 $O/pytest_ops.log $O/pytest_model.log $O/pytest_full.log
cat $O/mb_fwd_tiled.log $O/mb_fwd_ws.log $O/mb_fused.log
for f in base new occ3 fusedonly; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; gemm frac", round(r.get("roofline",{}).get("frac",0),4), "final_loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
