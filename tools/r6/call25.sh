#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c25; mkdir -p $out; cd $REPO
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d.get('by_batch'))"
DL3_NARROW=0 python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('narrow off', d['value'], d['ms_per_step'], d.get('by_batch'))"
