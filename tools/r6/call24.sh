#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c24; mkdir -p $out; cd $REPO
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "logits or narrow or (test_pwconv_fwd and not weight) or test_pwconv_bwd_weight or test_pwconv_bwd_data" 2>&1 | tail -15
S="lfwd:524288x256x21 lbwd:524288x256x21 lwgrad:524288x256x21"
echo "## default"; python tools/r6/gemm_bench.py $S
echo "## off"; DL3_NARROW=0 python tools/r6/gemm_bench.py $S
echo "## B=16"; python tools/r6/gemm_bench.py lfwd:65536x256x21 lbwd:65536x256x21 lwgrad:65536x256x21
echo "## B=16 off"; DL3_NARROW=0 python tools/r6/gemm_bench.py lfwd:65536x256x21 lbwd:65536x256x21 lwgrad:65536x256x21
echo "## B=2"; python tools/r6/gemm_bench.py lfwd:8192x256x21 lbwd:8192x256x21 lwgrad:8192x256x21
echo "## B=2 off"; DL3_NARROW=0 python tools/r6/gemm_bench.py lfwd:8192x256x21 lbwd:8192x256x21 lwgrad:8192x256x21
