#!/bin/bash
# round 4, GPU call 21: fused both-gradient kernel at small batches — minimum row count for choosing it (DL3_FUSED_ROWS, default 131072)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4u; mkdir -p $out
cd $REPO
bash tools/r4/ab.sh r4u/ab2 "--steps 100 --warmup 3 --batch 2" "1_r131072|DL3_DY_MAT=1" "2_r32768|DL3_FUSED_ROWS=32768" "3_r8192|DL3_FUSED_ROWS=8192" "4_r131072_again|DL3_DY_MAT=1" "5_r32768_again|DL3_FUSED_ROWS=32768" | tee $out/ab2.txt
bash tools/r4/ab.sh r4u/ab4 "--steps 80 --warmup 3 --batch 4" "1_r131072|DL3_DY_MAT=1" "2_r32768|DL3_FUSED_ROWS=32768" "3_r65536|DL3_FUSED_ROWS=65536" | tee $out/ab4.txt
bash tools/r4/ab.sh r4u/ab8 "--steps 60 --warmup 3 --batch 8" "1_r131072|DL3_DY_MAT=1" "2_r65536|DL3_FUSED_ROWS=65536" | tee $out/ab8.txt
