#!/bin/bash
# round 5, GPU call 2: static store counts (no store drain per tile / stage), benchmarked-plan parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
O=gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary or fused or pwconv_fwd" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
DL3_WS_VAR=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary and not full_size" > $O/pytest_ops_alt.log 2>&1
echo "ops alt rc=$?" >> $O/status.txt
timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_ws.log 2>&1
DL3_WS_VAR=1 timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_ws_alt.log 2>&1
timeout 600 python tools/r5/pw_hbm_bench.py fused > $O/mb_fused.log 2>&1
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
timeout 300 python bench.py $B --plan-json $O/plan_new.json > $O/bench_new.json 2> $O/bench_new.err
DL3_WS_VAR=1 timeout 300 python bench.py $B > $O/bench_alt.json 2> $O/bench_alt.err
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "notebook" > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "benchmarked_plan or b16" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
cat $O/status.txt
for f in $O/pytest_ops.log $O/pytest_ops_alt.log $O/pytest_model.log $O/pytest_full.log; do echo "== $f"; tail -n 25 $f; done
cat $O/mb_fwd_ws.log $O/mb_fwd_ws_alt.log $O/mb_fused.log
for f in new alt; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; gemm frac", round(r.get("roofline",{}).get("frac",0),4), "final_loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
