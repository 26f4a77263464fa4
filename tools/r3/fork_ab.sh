#!/bin/bash
# same-call A/B of the backward fork modes (DL3_FORK=0 | 2) on the headline config; writes gpurun_out/r3f/
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r3f; mkdir -p $out
cd $REPO
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "fork" > $out/t.log 2>&1; echo "test rc $?" >> $out/t.log
for rep in 1 2; do
  for f in 0 2; do
    DL3_FORK=$f timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline > $out/b_f${f}_$rep.json 2> $out/b_f${f}_$rep.err
  done
done
python - <<'PY' > $out/summary.txt
import json, glob, os
for p in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r3f/b_*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(os.path.basename(p), d["value"], d["ms_per_step"], d["config"].get("backward_fork"))
    except Exception as e:
        print(os.path.basename(p), "failed", e)
PY
cat $out/t.log | tail -5; cat $out/summary.txt
