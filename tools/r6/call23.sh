#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c23; mkdir -p $out; cd $REPO
S="lfwd:524288x256x21 lbwd:524288x256x21 lwgrad:524288x256x21"
echo "## default"; python tools/r6/gemm_bench.py $S
for c in 0 1 4 5 6 7 9; do echo "## wgrad cfg $c"; DL3_WGRAD_CFG=$c python tools/r6/gemm_bench.py lwgrad:524288x256x21; done
for c in 0 1 2 3 4 5 6; do echo "## gemm cfg $c"; DL3_GEMM_CFG=$c python tools/r6/gemm_bench.py lfwd:524288x256x21 lbwd:524288x256x21; done
