#!/bin/bash
# round 4, GPU call 22: remaining knobs at B = 2 (depthwise workgroup target, fold batching) — sweeps only
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4v; mkdir -p $out
cd $REPO
bash tools/r4/ab.sh r4v/ab2 "--steps 100 --warmup 3 --batch 2" "1_base|DL3_DY_MAT=1" "2_dw512|DL3_DW_BLOCKS=512" "3_dw2048|DL3_DW_BLOCKS=2048" "4_defer64|DL3_FOLD_DEFER_MB=64" "5_nobatchfold|DL3_BATCH_FOLDS=0" "6_dymatk32|DL3_DY_MAT_K=32" "7_base_again|DL3_DY_MAT=1" | tee $out/ab2.txt
bash tools/r4/ab.sh r4v/ab16 "--steps 40 --warmup 3 --batch 16" "1_base|DL3_DY_MAT=1" "2_dw2048|DL3_DW_BLOCKS=2048" "3_defer64|DL3_FOLD_DEFER_MB=64" "4_dymatk32|DL3_DY_MAT_K=32" | tee $out/ab16.txt
