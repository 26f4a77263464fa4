#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="bwd1:524288x960x160 bwd1:524288x576x96 bwd1:524288x384x64 bwd2:524288x64x384"
for i in 1 2; do
echo "## head"; DL3_LIBPATH=$REPO/build_variants/libdl3_head.so python tools/r6/gemm_bench.py $S
echo "## tree (laundered, xpre tn<=4)"; python tools/r6/gemm_bench.py $S
echo "## x5 (laundered, xpre tn<=5)"; DL3_LIBPATH=$REPO/build_variants/libdl3_x5.so python tools/r6/gemm_bench.py $S
done
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary_short" 2>&1 | tail -3
DL3_LIBPATH=$REPO/build_variants/libdl3_x5.so timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary_short" 2>&1 | tail -3
