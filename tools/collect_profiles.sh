#!/bin/bash
# Collects the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run (hipGraph replay)  -> per-kernel time table
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of the eager step -> HBM bytes
#      per step of the depthwise march family (tools/pmc_family.py)
# usage: bash tools/collect_profiles.sh <tag> <batch> [extra bench.py flags...]
# Outputs land in gpurun_out/prof_<tag>_*/ ; copy the summaries under profiles/.
set -u
TAG=${1:-r02}
B=${2:-64}
shift 2 || true
EXTRA="$@"
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_bench -o bench -- python $REPO/bench.py --batch $B --no-cpu-baseline --no-split-leg --steps 10 --warmup 3 --plan-json $REPO/gpurun_out/prof_${TAG}_plan.json $EXTRA > $REPO/gpurun_out/prof_${TAG}_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/gpurun_out/prof_${TAG}_pmc_$c -o pmc -- python $REPO/bench.py --batch $B --no-cpu-baseline --no-split-leg --no-roofline --no-graph --steps 3 --warmup 2 $EXTRA > $REPO/gpurun_out/prof_${TAG}_pmc_$c.log 2>&1
done
cd $REPO
find gpurun_out/prof_${TAG}_bench -name "*kernel_stats.csv" | head -2
tail -1 gpurun_out/prof_${TAG}_bench.log | cut -c1-300
