#!/bin/bash
out=gpurun_out/r3y; mkdir -p $out
export PROBE_SHAPES=4096x160x960,4096x960x160,4096x960x320,4096x64x384,4096x384x64
echo "== f32"; python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids | tee $out/phase_f32.log
echo "== split"; PROBE_MATH=split python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids | tee $out/phase_split.log
