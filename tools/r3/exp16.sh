#!/bin/bash
out=gpurun_out/r3p; mkdir -p $out
python tools/r3/shape_ab.py 2>&1 | grep -v amdgpu.ids | tee $out/shape_ab.log
