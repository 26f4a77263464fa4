#!/bin/bash
# round 4: the driver's own sequence on the final tree — the -m gpu suite, smoke(), the default bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4z; mkdir -p $out
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4z/bench_default.json").read().strip().splitlines()[-1])
print("value %.1f img/s  %.2f ms/step  gemm frac %.3f  atrous frac %.3f  split %.1f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_hbm"]["frac"], d["split_math"]["value"]))
print("by_batch", {k: round(v["value"], 1) for k, v in d["by_batch"].items()}, "configs", {k: round(v["value"], 1) for k, v in d["configs"].items()})
PY
