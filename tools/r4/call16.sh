#!/bin/bash
# round 4, GPU call 16: weight-gradient launches — target workgroup count (slab partial traffic vs fill) at B = 16 / 2 / 128
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4p; mkdir -p $out
cd $REPO
bash tools/r4/ab.sh r4p/ab16 "--steps 40 --warmup 3 --batch 16" "1_w1024|DL3_DY_MAT=1" "2_w512|DL3_WGRAD_WGS=512" "3_w768|DL3_WGRAD_WGS=768" "4_w1024_again|DL3_DY_MAT=1" "5_w512_again|DL3_WGRAD_WGS=512" "6_w256|DL3_WGRAD_WGS=256" | tee $out/ab16.txt
bash tools/r4/ab.sh r4p/ab2 "--steps 100 --warmup 3 --batch 2" "1_w1024|DL3_DY_MAT=1" "2_w512|DL3_WGRAD_WGS=512" "3_w2048|DL3_WGRAD_WGS=2048" | tee $out/ab2.txt
bash tools/r4/ab.sh r4p/ab128 "--steps 12 --warmup 3" "1_w1024|DL3_DY_MAT=1" "2_w512|DL3_WGRAD_WGS=512" "3_w768|DL3_WGRAD_WGS=768" | tee $out/ab128.txt
