#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c27; mkdir -p $out; cd $REPO
for i in 1 2; do
python bench.py --backbone xception --os 8 --batch 16 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-legs --no-split-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
DL3_COLSPLIT=0 python bench.py --backbone xception --os 8 --batch 16 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-legs --no-split-leg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('colsplit off', d['value'], d['ms_per_step'])"
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "cfg4 or xception" 2>&1 | tail -5
