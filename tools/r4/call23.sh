#!/bin/bash
# round 4, GPU call 23: rocprofv3 kernel stats of the by_batch legs' configurations (cfg2 at B = 2 and 16) on the final tree
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4w; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
prof() { tag=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o bench -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --plan-json $out/plan_$tag.json "$@" > $out/prof_$tag.log 2>&1; echo "$tag rc $?"; f=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" $out/${tag}_kernel_stats.csv; tail -1 $out/prof_$tag.log | cut -c1-160; }
prof cfg2_b2 --batch 2 --steps 50 --warmup 3
prof cfg2_b16 --batch 16 --steps 20 --warmup 3
rm -rf $out/prof_cfg2_b2 $out/prof_cfg2_b16
ls -la $out
