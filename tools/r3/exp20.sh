#!/bin/bash
out=gpurun_out/r3t; mkdir -p $out
export PROBE_SHAPES=4096x160x960,4096x960x160,4096x64x384
echo "== 512 workgroups (2 per CU)"; PROBE_KINDS=fwd,dgrad python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids
echo "== 256 workgroups (1 per CU)"; DL3_GEMM_PY=256 PROBE_KINDS=fwd,dgrad python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids
