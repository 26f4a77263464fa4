#!/bin/bash
# Collects the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench.py run           -> per-kernel time table
#   2. rocprofv3 --kernel-trace --stats of bench.py --roofline-only           -> average launch time of the roofline kernels
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/dw_only.py -> HBM bytes per launch
# Outputs land in gpurun_out/prof_<tag>/ ; tools/pmc_parse.py + the copy step put the summaries under profiles/.
set -u
TAG=${1:-r01}
B=${2:-64}
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_bench -o bench -- python $REPO/bench.py --batch $B --no-cpu-baseline --steps 10 --warmup 3 > $REPO/gpurun_out/prof_${TAG}_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG}_roofline -o roofline -- python $REPO/bench.py --batch $B --roofline-only > $REPO/gpurun_out/prof_${TAG}_roofline.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $REPO/gpurun_out/prof_${TAG}_pmc_$c -o pmc -- python $REPO/tools/dw_only.py $B > $REPO/gpurun_out/prof_${TAG}_pmc_$c.log 2>&1
done
cd $REPO
python tools/pmc_parse.py gpurun_out/prof_${TAG}_pmc_FETCH_SIZE gpurun_out/prof_${TAG}_pmc_WRITE_SIZE > gpurun_out/prof_${TAG}_pmc_raw.json
find gpurun_out/prof_${TAG}_bench gpurun_out/prof_${TAG}_roofline -name "*kernel_stats.csv" | head
tail -2 gpurun_out/prof_${TAG}_bench.log | cut -c1-400
tail -1 gpurun_out/prof_${TAG}_roofline.log
cat gpurun_out/prof_${TAG}_pmc_raw.json
