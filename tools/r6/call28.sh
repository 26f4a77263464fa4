#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
S="fwd:524288x64x384 fwd:524288x96x576"
for x in 0 1 2 3; do echo "## X4=$x"; DL3_WS2_X4=$x python tools/r6/gemm_bench.py $S; done
