#!/bin/bash
out=gpurun_out/r3l; mkdir -p $out
python tools/r3/phase_probe.py 128 2>&1 | grep -v amdgpu.ids > $out/phase_wait.log; cat $out/phase_wait.log
