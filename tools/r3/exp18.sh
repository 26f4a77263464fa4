#!/bin/bash
out=gpurun_out/r3r; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -m gpu -k "phase_shift or split_math_under or subpixel" 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
$B --head subpixel --no-split-leg 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('subpixel', round(r['value'],1), round(r['ms_per_step'],2))"
$B 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('cfg2', round(r['value'],1), 'split', r.get('split_math'))" | cut -c1-300
