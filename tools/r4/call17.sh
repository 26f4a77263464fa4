#!/bin/bash
# round 4, GPU call 17: weight-gradient M split by slab traffic (new default) against the fixed 1024-workgroup target, B = 16 / 32 / 64 / 4
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4q; mkdir -p $out
cd $REPO
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "bwd_weight or fused" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -1 $out/pytest_ops.log
bash tools/r4/ab.sh r4q/ab16 "--steps 40 --warmup 3 --batch 16" "1_fixed1024|DL3_WGRAD_WGS=1024" "2_rule|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4q/ab32 "--steps 30 --warmup 3 --batch 32" "1_fixed1024|DL3_WGRAD_WGS=1024" "2_rule|DL3_DY_MAT=1" "3_fixed1024_again|DL3_WGRAD_WGS=1024" "4_rule_again|DL3_DY_MAT=1" | tee $out/ab32.txt
bash tools/r4/ab.sh r4q/ab64 "--steps 20 --warmup 3 --batch 64" "1_fixed1024|DL3_WGRAD_WGS=1024" "2_rule|DL3_DY_MAT=1" | tee $out/ab64.txt
bash tools/r4/ab.sh r4q/ab4 "--steps 80 --warmup 3 --batch 4" "1_fixed1024|DL3_WGRAD_WGS=1024" "2_rule|DL3_DY_MAT=1" | tee $out/ab4.txt
bash tools/r4/ab.sh r4q/abx "--steps 6 --warmup 3 --batch 16 --backbone xception --os 8" "1_fixed1024|DL3_WGRAD_WGS=1024" "2_rule|DL3_DY_MAT=1" | tee $out/abx.txt
