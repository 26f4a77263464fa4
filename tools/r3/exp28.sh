#!/bin/bash
out=gpurun_out/r3aa; mkdir -p $out
export PROBE_SHAPES=4096x160x960,4096x960x160,4096x960x320,4096x64x384,4096x384x64,4096x96x576 PROBE_KINDS=wgrad
python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids | tee $out/phase_wgrad.log
