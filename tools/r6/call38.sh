#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "without_lds_stage" 2>&1 | grep -v "^$" | tail -40
S="wgraddy:524288x160x960 wgraddy:524288x960x160 wgraddy:524288x960x320 wgraddy:524288x576x160 wgrad:524288x960x160"
for i in 1 2; do
echo "## direct"; python tools/r6/gemm_bench.py $S
echo "## tiled"; DL3_WGRAD_DIRECT=0 python tools/r6/gemm_bench.py $S
done
