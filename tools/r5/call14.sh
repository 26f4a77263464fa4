#!/bin/bash
# round 5, GPU call 14: pipelined masked epilogue of the single-tensor bwd-data stream kernels (A/B against the committed library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c14
export TMPDIR=/tmp
O=gpurun_out/c14
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pwconv_bwd_data or pwconv" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline"
for rep in 1 2; do
DL3_LIBPATH=$PWD/build_variants/libdl3_c13.so timeout 300 python bench.py $B --plan-json $O/plan_old$rep.json > $O/bench_old$rep.json 2> $O/bench_old$rep.err
timeout 300 python bench.py $B --plan-json $O/plan_new$rep.json > $O/bench_new$rep.json 2> $O/bench_new$rep.err
done
cat $O/status.txt; tail -n 3 $O/pytest_ops.log
for f in old1 new1 old2 new2; do python - <<PY
import json
r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
rows=json.load(open("$O/plan_$f.json"))["rows"]
bd=sum(x["ms"] for x in rows if x["family"]=="gemm" and x["shape"].startswith("bwd-data"))
print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],3), "ms; gemm", round((r.get("roofline") or {}).get("frac",0),4), "bwd-data ms", round(bd,2), "loss", r["config"]["final_loss"])
PY
done
python - <<'PY'
import json
a=json.load(open("gpurun_out/c14/plan_old1.json"))["rows"]; b=json.load(open("gpurun_out/c14/plan_new1.json"))["rows"]
for x,y in zip(a,b):
    if x["family"]=="gemm" and x["shape"].startswith("bwd-data") and x["ms"]>0.2:
        print("%-44s old %.3f new %.3f" % (x["shape"][:44], x["ms"], y["ms"]))
PY
