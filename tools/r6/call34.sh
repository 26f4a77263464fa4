#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="fwd:65536x160x960 fwd:65536x96x576 fwd:65536x64x384 bwd1:65536x960x160 bwd1:65536x576x96 bwd1:65536x384x64 bwd1:65536x576x160"
for i in 1 2; do
echo "## default (tiled below 131072 rows)"; python tools/r6/gemm_bench.py $S
echo "## ws2 from 65536 rows"; DL3_WS2_MINROWS=65536 python tools/r6/gemm_bench.py $S
done
