#!/bin/bash
# round 4, GPU call 18: the narrowed weight-gradient split rule (160-wide tiles, <= 160x960, >= 32k rows) on / off
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4r; mkdir -p $out
cd $REPO
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "bwd_weight" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -1 $out/pytest_ops.log
bash tools/r4/ab.sh r4r/ab16 "--steps 40 --warmup 3 --batch 16" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" "3_off_again|DL3_WGRAD_HALVE=0" "4_on_again|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4r/ab32 "--steps 30 --warmup 3 --batch 32" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" | tee $out/ab32.txt
bash tools/r4/ab.sh r4r/ab8 "--steps 60 --warmup 3 --batch 8" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" | tee $out/ab8.txt
bash tools/r4/ab.sh r4r/ab4 "--steps 80 --warmup 3 --batch 4" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" | tee $out/ab4.txt
bash tools/r4/ab.sh r4r/abx "--steps 6 --warmup 3 --batch 16 --backbone xception --os 8" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" | tee $out/abx.txt
bash tools/r4/ab.sh r4r/ab128 "--steps 12 --warmup 3" "1_off|DL3_WGRAD_HALVE=0" "2_on|DL3_DY_MAT=1" | tee $out/ab128.txt
