#!/bin/bash
# round 5, GPU call 3: new host-side paths (DP arena tail, feeder, trainable snapshot), benchmarked-plan parity, same-call A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
O=gpurun_out/c3
timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q -m gpu -s > $O/pytest_parallel.log 2>&1
echo "parallel rc=$?" > $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "feeder or notebook or fit_generator or adam or optimizer" > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "benchmarked_plan or b16" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
timeout 600 python tools/r5/xception_layer_distance.py > $O/xception_layer_distance.txt 2> $O/xception_layer_distance.err
B="--steps 10 --warmup 3 --no-legs --no-split-leg --no-cpu-baseline --no-roofline"
for rep in 1 2; do
DL3_FWD_WS=0 DL3_FUSED_V=1 timeout 300 python bench.py $B > $O/bench_base$rep.json 2> $O/bench_base$rep.err
timeout 300 python bench.py $B > $O/bench_new$rep.json 2> $O/bench_new$rep.err
DL3_WS_VAR=1 timeout 300 python bench.py $B > $O/bench_alt$rep.json 2> $O/bench_alt$rep.err
done
REPS=12 timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_ws.log 2>&1
REPS=12 DL3_WS_VAR=1 timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_ws_alt.log 2>&1
REPS=12 DL3_FWD_WS=0 timeout 300 python tools/r5/pw_hbm_bench.py fwd > $O/mb_fwd_tiled.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --plan-json $O/plan_default.json > $O/bench_default.json 2> $O/bench_default.err
cat $O/status.txt
for f in $O/pytest_parallel.log $O/pytest_model.log $O/pytest_full.log; do echo "== $f"; tail -n 30 $f; done
echo "== xception"; tail -n 60 $O/xception_layer_distance.txt; tail -n 5 $O/xception_layer_distance.err
cat $O/mb_fwd_tiled.log $O/mb_fwd_ws.log $O/mb_fwd_ws_alt.log
for f in base1 new1 alt1 base2 new2 alt2 default; do python - <<PY
import json
try:
    r=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(r["value"],1), "img/s", round(r["ms_per_step"],2), "ms; final_loss", r["config"]["final_loss"], "sync guard", r["config"].get("host_syncs_in_timed_loop"))
    if "by_batch" in r:
        for b,v in r["by_batch"].items(): print("  B=%s %.1f img/s  fed %s" % (b, v["value"], v.get("fed",{}).get("vs_resident")))
        print("  fed128", r.get("fed",{}).get("vs_resident"), r.get("fed",{}).get("value"))
        for k,v in r["configs"].items(): print("  ", k, round(v["value"],1))
        print("  gemm frac", r["roofline"]["frac"], "hbm frac", r["roofline_hbm"]["frac"])
except Exception as e:
    print("$f failed", e)
PY
done
tail -n 12 $O/bench_default.err
