#!/bin/bash
REPO=$(pwd); out=$REPO/gpurun_out/forktrace; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for f in 0 1; do
DL3_FORK=$f timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/f$f -o t -- python $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-split-leg --no-roofline > $out/f$f.log 2>&1
done
cd $REPO; python tools/r3/fork_trace.py $out/f0 $out/f1 > $out/summary.json; cat $out/summary.json; rm -rf $out/f0 $out/f1
