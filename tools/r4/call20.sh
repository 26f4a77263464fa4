#!/bin/bash
# round 4, GPU call 20: fused both-gradient kernel — minimum rows per workgroup (slab overhead at few rows)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4t; mkdir -p $out
cd $REPO
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fused" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -1 $out/pytest_ops.log
bash tools/r4/ab.sh r4t/ab16 "--steps 40 --warmup 3 --batch 16" "1_old2048|DL3_FUSED_MINROWS=32" "2_min256|DL3_FUSED_MINROWS=256" "3_min512|DL3_DY_MAT=1" "4_min1024|DL3_FUSED_MINROWS=1024" "5_old_again|DL3_FUSED_MINROWS=32" "6_min512_again|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4t/ab32 "--steps 30 --warmup 3 --batch 32" "1_old2048|DL3_FUSED_MINROWS=32" "2_min512|DL3_DY_MAT=1" "3_min1024|DL3_FUSED_MINROWS=1024" | tee $out/ab32.txt
bash tools/r4/ab.sh r4t/ab2 "--steps 100 --warmup 3 --batch 2" "1_old2048|DL3_FUSED_MINROWS=32" "2_min512|DL3_DY_MAT=1" | tee $out/ab2.txt
bash tools/r4/ab.sh r4t/ab128 "--steps 12 --warmup 3" "1_old2048|DL3_FUSED_MINROWS=32" "2_min512|DL3_DY_MAT=1" "3_min1024|DL3_FUSED_MINROWS=1024" | tee $out/ab128.txt
