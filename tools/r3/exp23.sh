#!/bin/bash
out=gpurun_out/r3w; mkdir -p $out
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-leg --backbone xception --os 8 --batch 16"
for blk in 2048 512 1024 1536 3072 4096; do
DL3_DW_BLOCKS=$blk $B 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('DW_BLOCKS $blk', round(r['value'],2), 'hbm frac', round(r['roofline_hbm']['frac'],4), 'dw ms', round(r['roofline_hbm']['family_ms_per_step'],3), 'fwd', round(r['roofline_hbm']['forward_only']['frac'],3))"
done
B2="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg"
for blk in 2048 1024 4096; do
DL3_DW_BLOCKS=$blk $B2 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cfg2 DW_BLOCKS $blk', round(r['value'],2), 'hbm frac', round(r['roofline_hbm']['frac'],4), 'dw ms', round(r['roofline_hbm']['family_ms_per_step'],3))"
done
