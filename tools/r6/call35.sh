#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="bwd1:32768x960x160 bwd1:32768x576x96 bwd1:32768x576x160 bwd1:98304x960x160 bwd1:98304x576x96 fwd:98304x160x960 fwd:98304x96x576"
for i in 1 2; do
echo "## default"; python tools/r6/gemm_bench.py $S
echo "## ws2 from 32768 rows"; DL3_WS2_MINROWS=32768 python tools/r6/gemm_bench.py $S
done
