#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c7; mkdir -p $out; cd $REPO
W="wgraddy:65536x736x736 wgraddy:65536x1536x1536 wgraddy:65536x736x1024"
for c in -1 0 1 2 3 4 5 8 9; do
  for wgs in 0 512 2048; do
    echo "## cfg $c wgs $wgs"; DL3_WGRAD_CFG=$c DL3_WGRAD_WGS=$wgs python tools/r6/gemm_bench.py $W
  done
done 2>&1 | grep -v amdgpu.ids | tee $out/wgrad_x.txt
