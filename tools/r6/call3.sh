#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c3; mkdir -p $out; cd $REPO
W="wgrad:65536x736x736 wgraddy:65536x736x736 wgraddy:524288x160x960 wgraddy:524288x960x160 wgraddy:524288x96x576 wgraddy:524288x576x96 wgraddy:524288x576x160 wgraddy:524288x960x320 wgraddy:524288x384x96 wgrad:524288x64x384 wgraddy:524288x384x64 wgraddy:524288x256x256"
for rep in 1 2; do
echo "## base lib"; DL3_LIBPATH=$REPO/build_variants/libdl3_base.so python tools/r6/gemm_bench.py $W
echo "## new lib (LDS coefficients)"; python tools/r6/gemm_bench.py $W
echo "## new lib (register coefficients)"; DL3_LIBPATH=$REPO/build_variants/libdl3_cfreg.so python tools/r6/gemm_bench.py $W
done 2>&1 | grep -v amdgpu.ids | tee $out/wgrad.txt
S="fwd:65536x736x736 bwd1:65536x736x736"
for v in "DL3_GEMM_JV=0" "DL3_JV_PF=128 DL3_JV_PL=128" "DL3_JV_PF=171 DL3_JV_PL=171" "DL3_JV_PF=256 DL3_JV_PL=256" "DL3_JV_PF=512 DL3_JV_PL=512" "DL3_GEMM_JV=0 DL3_GEMM_PY=2560" "DL3_GEMM_JV=0 DL3_GEMM_PY=1280"; do
  echo "## $v"; env $v python tools/r6/gemm_bench.py $S
done 2>&1 | grep -v amdgpu.ids | tee $out/jv.txt
