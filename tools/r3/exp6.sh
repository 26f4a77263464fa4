#!/bin/bash
out=gpurun_out/r3f; mkdir -p $out
python tools/r3/phase_probe.py 128 > $out/phase_b128.log 2>&1; cat $out/phase_b128.log
