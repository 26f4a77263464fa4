#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="wgraddy:65536x736x736 wgraddy:524288x160x960 wgraddy:524288x96x576 wgraddy:524288x576x96 wgraddy:524288x960x160 wgraddy:524288x960x320 wgraddy:524288x320x256 wgraddy:524288x384x64 wgraddy:65536x1536x1536"
for i in 1 2; do
echo "## MS=16 (tree)"; python tools/r6/gemm_bench.py $S
echo "## MS=32"; DL3_LIBPATH=$REPO/build_variants/libdl3_ms32.so python tools/r6/gemm_bench.py $S
done
