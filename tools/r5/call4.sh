#!/bin/bash
# round 5, GPU call 4: wide dX stores through LDS (fused whole-block shapes), double-precision pooled branch, fixed tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
O=gpurun_out/c4
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary or fused or pwconv_fwd or few_rows or gap" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q -m gpu -s > $O/pytest_parallel.log 2>&1
echo "parallel rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg4_xception_os8_512_forward or (cfg2_cfg3 and deeplab)" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
timeout 600 python tools/r5/xception_layer_distance.py > $O/xception_layer_distance.txt 2> $O/xception_layer_distance.err
REPS=12 timeout 600 python tools/r5/pw_hbm_bench.py fused > $O/mb_fused.log 2>&1
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --no-roofline"
for rep in 1 2; do
DL3_FWD_WS=0 DL3_FUSED_V=1 timeout 300 python bench.py $B > $O/bench_base$rep.json 2> $O/bench_base$rep.err
timeout 300 python bench.py $B > $O/bench_new$rep.json 2> $O/bench_new$rep.err
DL3_FUSED_TILE=0 timeout 300 python bench.py $B > $O/bench_notile$rep.json 2> $O/bench_notile$rep.err
done
cat $O/status.txt
for f in $O/pytest_ops.log $O/pytest_parallel.log $O/pytest_model.log $O/pytest_full.log; do echo "== $f"; tail -n 25 $f; done
echo "== xception"; tail -n 16 $O/xception_layer_distance.txt; tail -n 3 $O/xception_layer_distance.err
cat $O/mb_fused.log
for f in base1 new1 notile1 base2 new2 notile2; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; final_loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
