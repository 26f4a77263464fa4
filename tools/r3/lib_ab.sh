#!/bin/bash
# same-call A/B of two builds of libdl3.so: build_variants/libdl3_base.so against the in-tree library, B = 128 / 16 / 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r3ab; mkdir -p $out; rm -f $out/b*.json
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > $out/t.log 2>&1; echo "ops rc $?" >> $out/t.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "${AB_TESTS:-poison or fork}" >> $out/t.log 2>&1; echo "model rc $?" >> $out/t.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
for rep in 1 2; do
  for b in 128 16 2; do
    DL3_LIBPATH=$REPO/build_variants/libdl3_base.so timeout 600 $B --batch $b > $out/b${b}_base_$rep.json 2> $out/b${b}_base_$rep.err
    timeout 600 $B --batch $b > $out/b${b}_new_$rep.json 2> $out/b${b}_new_$rep.err
  done
done
python - <<'PY' > $out/summary.txt
import json, glob, os
for p in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r3ab/b*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(os.path.basename(p), round(d["value"], 1), round(d["ms_per_step"], 3))
    except Exception as e:
        print(os.path.basename(p), "failed", e)
PY
grep -E 'passed|failed|rc' $out/t.log; cat $out/summary.txt
