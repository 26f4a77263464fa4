#!/bin/bash
# round 5, GPU call 5: depthwise march kernels without the early break (no store drain in front of each row group),
# concat_projection mean in the per-image GEMM, frozen-BN gradient diagnostics
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
O=gpurun_out/c5
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv or fused or few_rows" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -s > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
DL3_ROWS_F64=0 timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "frozen_bn_train_step" > $O/pytest_model_norows.log 2>&1
echo "model(no f64 rows) rc=$?" >> $O/status.txt
DL3_LIBPATH=$PWD/build_variants/libdl3_dwold.so timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "frozen_bn_train_step" > $O/pytest_model_dwold.log 2>&1
echo "model(old dw) rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg4_xception_os8_512_forward" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
timeout 600 python tools/r5/xception_layer_distance.py > $O/xception_layer_distance.txt 2> $O/xception_layer_distance.err
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_dwold.so timeout 300 python bench.py $B --plan-json $O/plan_dwold$rep.json > $O/bench_dwold$rep.json 2> $O/bench_dwold$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
DL3_LIBPATH=$PWD/build_variants/libdl3_dwold.so timeout 300 python bench.py --steps 5 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --backbone xception --os 8 --batch 16 > $O/bench_x_dwold.json 2> $O/bench_x_dwold.err
timeout 300 python bench.py --steps 5 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --backbone xception --os 8 --batch 16 > $O/bench_x_new.json 2> $O/bench_x_new.err
cat $O/status.txt
for f in $O/pytest_ops.log $O/pytest_full.log; do echo "== $f"; tail -n 8 $f; done
echo "== model"; grep -n "frozen-BN\|share\|passed\|failed\|FAILED\|Error" $O/pytest_model.log | head -40
echo "== model no rows f64"; grep -n "frozen-BN\|share\|passed\|failed" $O/pytest_model_norows.log | head
echo "== model old dw"; grep -n "frozen-BN\|share\|passed\|failed" $O/pytest_model_dwold.log | head
echo "== xception"; tail -n 12 $O/xception_layer_distance.txt
for f in dwold1 new1 dwold2 new2 x_dwold x_new; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; gemm", round(r.get("roofline",{}).get("frac",0),4), "atrous", round(r.get("roofline_hbm",{}).get("frac",0),4), "loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
