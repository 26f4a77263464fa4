#!/bin/bash
# same-call A/B of environment / library variants on one bench configuration (GPU box):
#   bash tools/r4/ab.sh OUTDIR "bench args" "NAME|ENV=V ENV2=V ..." ...
# every variant runs `python bench.py <bench args> --no-cpu-baseline --no-split-leg --no-legs --plan-json OUT/NAME.plan.json`
# and leaves OUT/NAME.json; a summary table (img/s, ms/step, per-direction GEMM milliseconds from the in-situ plan) is printed
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/$1; shift
bargs=$1; shift
mkdir -p $out
cd $REPO
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  env $envs timeout 900 python bench.py $bargs --no-cpu-baseline --no-split-leg --no-legs --plan-json $out/$name.plan.json \
      > $out/$name.json 2> $out/$name.err
done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
print("%-28s %9s %9s | %8s %8s %8s %8s %8s %8s" % ("variant", "img/s", "ms/step", "gemm fwd", "bwd-data", "bwd-wgt", "bwd-fusd", "dw", "other"))
for p in sorted(glob.glob(os.path.join(out, "*.json"))):
    if p.endswith(".plan.json"):
        continue
    name = os.path.basename(p)[:-5]
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print("%-28s failed: %s" % (name, e))
        continue
    fam = {"fwd": 0.0, "bwd-data": 0.0, "bwd-weight": 0.0, "bwd-fused": 0.0, "dw": 0.0, "other": 0.0}
    try:
        for r in json.load(open(os.path.join(out, name + ".plan.json")))["rows"]:
            if r["family"] == "gemm":
                k = r["shape"].split(" ")[0]
                fam[k if k in fam else "bwd-weight"] += r["ms"]
            elif r["family"].startswith("dw"):
                fam["dw"] += r["ms"]
            else:
                fam["other"] += r["ms"]
    except Exception:
        pass
    print("%-28s %9.1f %9.3f | %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f" % (name, d["value"], d["ms_per_step"], fam["fwd"], fam["bwd-data"],
                                                                   fam["bwd-weight"], fam["bwd-fused"], fam["dw"], fam["other"]))
PY
