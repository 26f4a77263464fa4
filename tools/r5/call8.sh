#!/bin/bash
# round 5, GPU call 8: depthwise march kernels with counted stores in interior row groups (A/B against the previous library),
# bit-identity of their outputs, stem digest check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
O=gpurun_out/c8
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv or conv3x3" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 200 python tools/r5/dw_bench.py > $O/dw_old.txt 2>&1
timeout 200 python tools/r5/dw_bench.py > $O/dw_new.txt 2>&1
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 100 python tools/r5/stem_bench.py > $O/stem_old.txt 2>&1
timeout 100 python tools/r5/stem_bench.py > $O/stem_new.txt 2>&1
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py $B > $O/bench_old$rep.json 2> $O/bench_old$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py --steps 6 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --backbone xception --os 8 --batch 16 > $O/bench_x_old.json 2> $O/bench_x_old.err
timeout 300 python bench.py --steps 6 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --backbone xception --os 8 --batch 16 > $O/bench_x_new.json 2> $O/bench_x_new.err
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
cat $O/status.txt
tail -n 4 $O/pytest_ops.log; tail -n 3 $O/pytest_model.log
echo "== dw old"; grep "^dw" $O/dw_old.txt; echo "== dw new"; grep "^dw" $O/dw_new.txt
grep stem $O/stem_old.txt $O/stem_new.txt
for f in old1 new1 old2 new2 x_old x_new; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; gemm", round(r.get("roofline",{}).get("frac",0),4), "atrous", round(r.get("roofline_hbm",{}).get("frac",0),4), "loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
