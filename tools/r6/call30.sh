#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="bwd1:524288x960x160 bwd1:524288x576x96 bwd1:524288x384x64 bwd1:524288x384x96 bwd2:524288x64x384"
for i in 1 2; do
echo "## shipped"; python tools/r6/gemm_bench.py $S
echo "## xpre"; DL3_LIBPATH=$REPO/build_variants/libdl3_xpre.so python tools/r6/gemm_bench.py $S
done
DL3_LIBPATH=$REPO/build_variants/libdl3_xpre.so timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary_short" 2>&1 | tail -3
