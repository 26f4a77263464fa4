#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c13; mkdir -p $out; cd $REPO
DL3_WS2=1 timeout 600 python -m pytest tests/gpu_ws2_probe.py -q -x 2>&1 | tail -8
S="bwd1:524288x960x160 bwd1:524288x576x96 bwd1:524288x384x96 fwd:524288x160x960"
for rep in 1 2; do
for v in "DL3_WS2=0" "DL3_WS2=1"; do echo "## $v"; env $v python tools/r6/gemm_bench.py $S; done
done 2>&1 | grep -v amdgpu.ids | tee $out/ws2.txt
