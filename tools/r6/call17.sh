#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c17; mkdir -p $out; cd $REPO
timeout 1500 python tools/r6/xception_argmax_stats.py 16 2>&1 | grep -v amdgpu.ids | tee $out/xception_argmax_16.txt
