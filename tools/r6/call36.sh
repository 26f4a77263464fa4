#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary_short" 2>&1 | tail -3
for i in 1 2; do
for b in 16 32; do
python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-split-leg --no-legs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree B$b', d['value'], d['ms_per_step'])"
DL3_LIBPATH=$REPO/build_variants/libdl3_head.so python bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-split-leg --no-legs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('head B$b', d['value'], d['ms_per_step'])"
done; done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
