#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c4; mkdir -p $out; cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "bwd_weight" > $out/ops.log 2>&1; echo "ops rc $?" >> $out/ops.log; tail -3 $out/ops.log
W="wgrad:65536x736x736 wgraddy:65536x736x736 wgraddy:524288x160x960 wgraddy:524288x960x160 wgraddy:524288x96x576 wgraddy:524288x576x96 wgraddy:524288x576x160 wgraddy:524288x960x320 wgraddy:524288x384x96 wgrad:524288x64x384 wgraddy:524288x384x64 wgraddy:524288x256x256"
for rep in 1 2; do
echo "## base lib"; DL3_LIBPATH=$REPO/build_variants/libdl3_base.so python tools/r6/gemm_bench.py $W
echo "## new lib"; python tools/r6/gemm_bench.py $W
done 2>&1 | grep -v amdgpu.ids | tee $out/wgrad.txt
