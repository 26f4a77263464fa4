#!/bin/bash
# round 4, GPU call 13: (a) prefetching bwd-data kernel at 128x160 tiles for 160-multiples (DL3_GEMM_PRE5=1; 200 B of scratch
# around the K loop), (b) fused kernel with the dW loop unrolled by 4 and three workgroups per CU for the five-block shapes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4m; mkdir -p $out
cd $REPO
V=$REPO/build_variants/libdl3_fused_u4.so
DL3_GEMM_PRE5=1 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "bwd_data" > $out/pytest_ops.log 2>&1; echo "ops(pre5) rc $?"; tail -1 $out/pytest_ops.log
DL3_LIBPATH=$V timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fused" > $out/pytest_ops2.log 2>&1; echo "ops(u4) rc $?"; tail -1 $out/pytest_ops2.log
bash tools/r4/ab.sh r4m/ab128 "--steps 15 --warmup 3" "1_base|DL3_DY_MAT=1" "2_pre5|DL3_GEMM_PRE5=1" "3_fused_u4|DL3_LIBPATH=$V" "4_base_again|DL3_DY_MAT=1" "5_pre5_again|DL3_GEMM_PRE5=1" "6_fused_u4_again|DL3_LIBPATH=$V" | tee $out/ab128.txt
