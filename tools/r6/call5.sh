#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c5; mkdir -p $out; cd $REPO
W=""
for s in 524288x160x960 524288x960x160 524288x96x576 524288x576x96 524288x576x160 524288x960x320 524288x384x96 524288x384x64 524288x256x256 524288x320x256; do W="$W wgrad:$s wgraddy:$s"; done
for rep in 1 2; do
echo "## base lib"; DL3_LIBPATH=$REPO/build_variants/libdl3_base.so python tools/r6/gemm_bench.py $W
done 2>&1 | grep -v amdgpu.ids | tee $out/wgrad_dy_cost.txt
