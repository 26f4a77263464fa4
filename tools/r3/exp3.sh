#!/bin/bash
out=gpurun_out/r3c; mkdir -p $out
python tools/r3/bn_probe.py 256 xception > $out/bn_probe_x256.log 2>&1; grep "rel err" $out/bn_probe_x256.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "bn_finalize" -s 2>&1 | tail -8
timeout 2400 python -m pytest tests/ -q -m gpu -x -s > $out/pytest_full.log 2>&1; echo "rc $?" >> $out/pytest_full.log
grep -n "argmax flips\|logits rel err\|whole vector\|error ratio\|passed\|failed\|rc " $out/pytest_full.log | tail -60
