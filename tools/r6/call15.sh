#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c15; mkdir -p $out; cd $REPO
S="fwd:524288x64x384 bwd1:524288x384x64 bwd2:524288x64x384 fwd:524288x384x64"
for rep in 1 2; do
for v in "DL3_WS2_K64=0" "DL3_WS2_K64=1"; do echo "## $v"; env $v python tools/r6/gemm_bench.py $S; done
done 2>&1 | grep -v amdgpu.ids | tee $out/ws2.txt
