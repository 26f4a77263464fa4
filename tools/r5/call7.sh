#!/bin/bash
# round 5, GPU call 7: the stem convolution's forward kernel pipelined around its stores (A/B against the previous library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
O=gpurun_out/c7
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 120 python tools/r5/stem_bench.py > $O/stem_old$rep.txt 2>&1
timeout 120 python tools/r5/stem_bench.py > $O/stem_new$rep.txt 2>&1
done
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg4_xception_os8_512_forward" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py $B > $O/bench_old$rep.json 2> $O/bench_old$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
cat $O/status.txt
tail -n 4 $O/pytest_ops.log; tail -n 8 $O/pytest_full.log; tail -n 3 $O/pytest_model.log
cat $O/stem_old1.txt $O/stem_new1.txt $O/stem_old2.txt $O/stem_new2.txt | grep stem
for f in old1 new1 old2 new2; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; gemm", round(r.get("roofline",{}).get("frac",0),4), "loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
