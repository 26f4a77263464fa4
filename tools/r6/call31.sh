#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "weight_stationary_short or single_tensor_masked or test_pwconv_bwd_data" 2>&1 | tail -3
python tools/r6/gemm_bench.py bwd1:524288x960x160 bwd1:524288x576x96 bwd1:524288x384x64 bwd1:524288x384x96 bwd2:524288x64x384
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-legs --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B128', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "benchmarked_plan_cfg2" 2>&1 | tail -3
