#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c26; mkdir -p $out; cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "column_split or long_reduction or test_pwconv_bwd_weight or single_tensor_masked" 2>&1 | tail -8
S="fwd:65536x736x736 bwd1:65536x736x736 bwd2:65536x736x736 fwd:65536x256x736 bwd1:65536x1024x736 fwd:65536x736x1024"
echo "## default"; python tools/r6/gemm_bench.py $S
echo "## off"; DL3_COLSPLIT=0 python tools/r6/gemm_bench.py $S
echo "## auto"; python tools/r6/gemm_bench.py fwd:65536x736x480 bwd1:65536x480x736 fwd:65536x736x256 bwd1:65536x256x736
