#!/bin/bash
out=gpurun_out/r3d; mkdir -p $out
python tools/gemm_tune.py --batch 128 --kinds fwd,dgrad > $out/tune_b128.log 2>&1
DL3_GEMM_PY=1024 python tools/gemm_tune.py --batch 128 --kinds fwd,dgrad --cfgs 4,7,8,9 > $out/tune_b128_py1024.log 2>&1
cat $out/tune_b128.log; echo ======= PY1024; cat $out/tune_b128_py1024.log
