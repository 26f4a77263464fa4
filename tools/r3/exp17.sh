#!/bin/bash
out=gpurun_out/r3q; mkdir -p $out
timeout 2400 python -m pytest tests/ -q -m gpu -x > $out/pytest_full.log 2>&1; echo "rc $?" >> $out/pytest_full.log
tail -8 $out/pytest_full.log
