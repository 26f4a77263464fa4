#!/bin/bash
out=gpurun_out/r3g; mkdir -p $out
python tools/gemm_tune.py --batch 128 --kinds fwd,dgrad --cfgs 3,4,7,8 > $out/tune_occ3_py512.log 2>&1
DL3_GEMM_PY=768 python tools/gemm_tune.py --batch 128 --kinds fwd,dgrad --cfgs 4,7,8 > $out/tune_occ3_py768.log 2>&1
cat $out/tune_occ3_py512.log; echo ===== PY768; cat $out/tune_occ3_py768.log
