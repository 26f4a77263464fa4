#!/bin/bash
out=gpurun_out/r3z; mkdir -p $out
DL3_GEMM_MATH=split timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "pwconv or gemm or pw_ or split" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -x -m gpu -k "split" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
for rep in 1 2; do
$B 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg2 f32', round(r['value'],1), 'split', round(r['split_math']['value'],1), round(r['split_math']['ms_per_step'],2))"
done
$B --backbone xception --os 8 --batch 16 --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg4 f32', round(r['value'],1), 'split', round(r['split_math']['value'],1))"
$B --head subpixel 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg3 f32', round(r['value'],1), 'split', round(r['split_math']['value'],1))"
