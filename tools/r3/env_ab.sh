#!/bin/bash
# same-call A/B of one environment toggle on the headline config at B = 128 / 16 / 2:
#   bash tools/r3/env_ab.sh DL3_BATCH_FOLDS 0 1     (writes gpurun_out/r3env/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; A=$2; Bv=$3
out=$REPO/gpurun_out/r3env; mkdir -p $out; rm -f $out/b*.json
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "${AB_OPS:-reduce or bwd_weight or bn}" > $out/t.log 2>&1; echo "ops rc $?" >> $out/t.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "${AB_TESTS:-poison or fork or batched}" >> $out/t.log 2>&1; echo "model rc $?" >> $out/t.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
for rep in 1 2; do
  for b in 128 16 2; do
    for v in $A $Bv; do
      env $VAR=$v timeout 600 $B --batch $b > $out/b${b}_${VAR}${v}_$rep.json 2> $out/b${b}_${VAR}${v}_$rep.err
    done
  done
done
python - <<'PY' > $out/summary.txt
import json, glob, os
for p in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r3env/b*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(os.path.basename(p), round(d["value"], 1), round(d["ms_per_step"], 3), d["config"].get("launches_per_step"))
    except Exception as e:
        print(os.path.basename(p), "failed", e)
PY
grep -E 'passed|failed|rc|Error' $out/t.log | tail -8; cat $out/summary.txt
