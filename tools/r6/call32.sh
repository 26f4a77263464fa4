#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="bwd1:524288x576x96 bwd1:524288x384x96 bwd1:524288x384x64"
for i in 1 2; do
echo "## head"; DL3_LIBPATH=$REPO/build_variants/libdl3_head.so python tools/r6/gemm_bench.py $S
echo "## xpre macro"; DL3_LIBPATH=$REPO/build_variants/libdl3_xpre.so python tools/r6/gemm_bench.py $S
echo "## tree"; python tools/r6/gemm_bench.py $S
done
