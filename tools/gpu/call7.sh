mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "conv3x3" > gpurun_out/r2g/pytest_conv.log 2>&1; tail -15 gpurun_out/r2g/pytest_conv.log
python -m pytest tests -m gpu -q -rf -k "not fullsize and not parallel" > gpurun_out/r2g/pytest_fast.log 2>&1; tail -5 gpurun_out/r2g/pytest_fast.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rf -k "xception" > gpurun_out/r2g/pytest_xc.log 2>&1; tail -5 gpurun_out/r2g/pytest_xc.log
python bench.py --backbone xception --os 8 --batch 16 --no-cpu-baseline --steps 8 --warmup 3 --plan-json gpurun_out/r2g/plan_xc8_b16.json > gpurun_out/r2g/cfg4_xception_os8_b16.json 2> gpurun_out/r2g/cfg4.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r2g/cfg4_xception_os8_b16.json')); print('cfg4', r['value'], r['ms_per_step'])
rows=json.load(open('gpurun_out/r2g/plan_xc8_b16.json'))['rows']
for x in rows:
    if 'conv3x3' in x['op']: print(x['op'], round(x['ms'],3))
PY
python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 | cut -c1-200
