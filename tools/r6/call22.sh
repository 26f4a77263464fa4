#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "bwd_weight" 2>&1 | tail -4
