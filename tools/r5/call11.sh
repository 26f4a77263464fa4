#!/bin/bash
# round 5, GPU call 11: contraction by the language rule in dwconv / conv3x3 — are the two bodies of the depthwise row loops
# bit-identical now (digests with DL3_DW_FAST=0 / 7), and the B=16 / B=2 engines of cfg4 with them?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
O=gpurun_out/c11
for f in 0 7; do DL3_DW_FAST=$f timeout 200 python tools/r5/dw_bench.py 16 > $O/dw16_fast$f.txt 2>&1; done
for f in 0 7; do DL3_DW_FAST=$f DL3_DW_TWO=0 timeout 200 python tools/r5/dw_bench.py 16 > $O/dw16_one_fast$f.txt 2>&1; done
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv or conv3x3" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "benchmarked_plan or cfg4_xception_os8_512_forward" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python bench.py $B > $O/bench_old$rep.json 2> $O/bench_old$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
cat $O/status.txt
tail -n 3 $O/pytest_ops.log; tail -n 3 $O/pytest_model.log; grep -n "against\|flips gpu\|passed\|failed\|assert " $O/pytest_full.log | tail -n 12
python - <<'PY'
import re
def rd(f):
    return [(m.group(1), m.group(2), m.group(3)) for m in (re.match(r"dw (\S+ r\d+) add=\d  fwd .*?\[(.*?)\] \| bwd .*?\[(.*?)\]", l) for l in open(f)) if m]
for tag in ("dw16", "dw16_one"):
    a, b = rd("gpurun_out/c11/%s_fast0.txt" % tag), rd("gpurun_out/c11/%s_fast7.txt" % tag)
    for x, y in zip(a, b):
        print(tag, x[0], "fwd", "same" if x[1] == y[1] else "DIFFERENT", "bwd", "same" if x[2] == y[2] else "DIFFERENT")
PY
for f in old1 new1 old2 new2; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],3), "ms; gemm", round((r.get("roofline") or {}).get("frac",0),4), "atrous", round((r.get("roofline_hbm") or {}).get("frac",0),4), "loss", r["config"]["final_loss"])
except Exception as e:
    print("$f failed", e)
PY
done
