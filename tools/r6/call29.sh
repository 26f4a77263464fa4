#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "## default"; python tools/r6/gemm_bench.py wgraddy:65536x736x736 wgraddy:65536x1536x2048 wgraddy:65536x2048x256
for c in 0 1 2 3 4 5 8 9; do echo "## wgrad cfg $c"; DL3_WGRAD_CFG=$c python tools/r6/gemm_bench.py wgraddy:65536x736x736; done
