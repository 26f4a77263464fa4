#!/bin/bash
out=gpurun_out/r3x; mkdir -p $out
python __graft_entry__.py smoke 2>&1 | tail -3
timeout 2400 python -m pytest tests/ -q -m gpu -x > $out/pytest_full.log 2>&1; echo "rc $?" >> $out/pytest_full.log
tail -4 $out/pytest_full.log
