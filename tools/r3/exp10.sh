#!/bin/bash
out=gpurun_out/r3j; mkdir -p $out
python tools/gemm_tune.py --batch 16 --kinds fwd,dgrad --cfgs 0,3 --shapes 4096x736x736,4096x736x768,4096x768x736,4096x768x768,4096x736x640,4096x640x736,4096x1024x1024,4096x704x704,4096x736x800 > $out/tune_736.log 2>&1
cat $out/tune_736.log
