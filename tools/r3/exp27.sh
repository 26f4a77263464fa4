#!/bin/bash
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
for rep in 1 2; do for blk in 1024 2048; do
DL3_DW_BLOCKS=$blk $B --backbone xception --os 16 --batch 16 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('x16 blocks $blk', round(r['value'],1))"
done; done
for blk in 1024 2048; do
DL3_DW_BLOCKS=$blk $B --batch 2 --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('b2 blocks $blk', round(r['value'],1))"
DL3_DW_BLOCKS=$blk $B --batch 16 --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('b16 blocks $blk', round(r['value'],1))"
done
