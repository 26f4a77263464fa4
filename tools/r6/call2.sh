#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c2; mkdir -p $out; cd $REPO
S="fwd:65536x736x736 bwd1:65536x736x736 bwd2:65536x736x736"
for v in "DL3_GEMM_JV=0" "X=1" "DL3_JV_PF=102 DL3_JV_PL=102" "DL3_JV_PF=102 DL3_JV_PL=61" "DL3_JV_PF=128 DL3_JV_PL=77" "DL3_JV_PF=222 DL3_JV_PL=136" "DL3_JV_PF=512 DL3_JV_PL=512"; do
  echo "## $v"; env $v python tools/r6/gemm_bench.py $S
done 2>&1 | tee $out/jv.txt
echo "## base lib"; DL3_LIBPATH=$REPO/build_variants/libdl3_base.so python tools/r6/gemm_bench.py $S wgrad:65536x736x736 wgraddy:65536x736x736 wgraddy:524288x160x960 wgraddy:524288x960x160 wgraddy:524288x96x576 wgrad:524288x64x384 2>&1 | tee $out/base.txt
echo "## new lib"; python tools/r6/gemm_bench.py wgrad:65536x736x736 wgraddy:65536x736x736 wgraddy:524288x160x960 wgraddy:524288x960x160 wgraddy:524288x96x576 wgrad:524288x64x384 2>&1 | tee $out/new.txt
