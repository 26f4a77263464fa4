mkdir -p gpurun_out/pre
python -m pytest tests/test_gpu_ops.py -q -k "pwconv" -x > gpurun_out/pre/pytest.log 2>&1; tail -5 gpurun_out/pre/pytest.log
for v in 0 1; do
  DL3_GEMM_PRE=$v python bench.py --no-cpu-baseline --steps 12 --warmup 4 --plan-json gpurun_out/pre/plan$v.json > gpurun_out/pre/b$v.json 2> gpurun_out/pre/b$v.err
  python -c "
import json;r=json.load(open('gpurun_out/pre/b$v.json'));print('pre $v', round(r['value'],1), r['roofline']['achieved'], r['roofline']['family_ms_per_step'])"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/pre/plan0.json'))['rows']; b=json.load(open('gpurun_out/pre/plan1.json'))['rows']
for x,y in zip(a,b):
    if x['op']=='dl3_pwconv_bwd_data' and abs(x['ms']-y['ms'])>0.03*x['ms']: print(x['shape'], round(x['ms'],3), '->', round(y['ms'],3))
PY
