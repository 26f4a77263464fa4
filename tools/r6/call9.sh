#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c9; mkdir -p $out; cd $REPO
for v in 0 1; do echo "## DL3_WS2=$v"; DL3_WS2=$v PROBE_KINDS=fwd PROBE_SHAPES=4096x160x960,4096x96x576,4096x960x160 python tools/r3/phase_probe.py 128; done 2>&1 | grep -v amdgpu.ids | tee $out/phase.txt
