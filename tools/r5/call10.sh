#!/bin/bash
# round 5, GPU call 10: depthwise fast-path rule (C % 32 == 0), stem weight gradient with 8 pixel pairs in flight; op + model
# + full-size tests on the new library, same-call bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
O=gpurun_out/c10
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv or conv3x3" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 100 python tools/r5/stem_bench.py > $O/stem_old.txt 2>&1
timeout 100 python tools/r5/stem_bench.py > $O/stem_new.txt 2>&1
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py $B > $O/bench_old$rep.json 2> $O/bench_old$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
for b in 2 16; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py --steps 20 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --no-roofline --batch $b > $O/bench_b${b}_old.json 2> $O/bench_b${b}_old.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --no-roofline --batch $b > $O/bench_b${b}_new.json 2> $O/bench_b${b}_new.err
done
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
cat $O/status.txt
tail -n 3 $O/pytest_ops.log; tail -n 3 $O/pytest_model.log; tail -n 5 $O/pytest_full.log
grep stem $O/stem_old.txt $O/stem_new.txt
for f in old1 new1 old2 new2 b2_old b2_new b16_old b16_new; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],3), "ms; gemm", round((r.get("roofline") or {}).get("frac",0),4), "atrous", round((r.get("roofline_hbm") or {}).get("frac",0),4), "loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
