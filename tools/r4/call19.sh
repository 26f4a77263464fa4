#!/bin/bash
# round 4, GPU call 19: knob sweeps at small batches — GEMM workgroup target (DL3_GEMM_PY), fused-kernel workgroups (DL3_FUSED_WGS)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4s; mkdir -p $out
cd $REPO
bash tools/r4/ab.sh r4s/ab16 "--steps 40 --warmup 3 --batch 16" "1_base|DL3_DY_MAT=1" "2_py512|DL3_GEMM_PY=512" "3_py1024|DL3_GEMM_PY=1024" "4_py4096|DL3_GEMM_PY=4096" "5_fw1024|DL3_FUSED_WGS=1024" "6_fw4096|DL3_FUSED_WGS=4096" "7_base_again|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4s/ab2 "--steps 100 --warmup 3 --batch 2" "1_base|DL3_DY_MAT=1" "2_py512|DL3_GEMM_PY=512" "3_py1024|DL3_GEMM_PY=1024" "4_py4096|DL3_GEMM_PY=4096" "5_fw512|DL3_FUSED_WGS=512" "6_nofused|DL3_FUSED_BWD=0" | tee $out/ab2.txt
