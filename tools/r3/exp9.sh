#!/bin/bash
out=gpurun_out/r3i; mkdir -p $out
python tools/gemm_tune.py --batch 16 --xception > $out/tune_x_b16.log 2>&1
cat $out/tune_x_b16.log
