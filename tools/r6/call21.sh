#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c21; mkdir -p $out; cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "bwd_data" 2>&1 | tail -4
S="bwd1:524288x160x960 bwd1:524288x960x320 bwd1:524288x96x576 bwd1:524288x576x160 bwd1:524288x320x256 bwd1:524288x256x256 bwd1:65536x160x960"
for rep in 1 2; do
for v in "DL3_GEMM_T16=0" "DL3_GEMM_T16=1"; do echo "## $v"; env $v python tools/r6/gemm_bench.py $S; done
done 2>&1 | grep -v amdgpu.ids | tee $out/t16.txt
