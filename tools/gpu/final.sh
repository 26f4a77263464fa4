mkdir -p gpurun_out/fin
python -m pytest tests -m gpu -q -rf > gpurun_out/fin/pytest_all.log 2>&1; tail -4 gpurun_out/fin/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
python bench.py --no-cpu-baseline --backbone xception --os 8 --batch 16 --steps 8 --warmup 3 > gpurun_out/fin/cfg4_xception_os8_b16.json 2> gpurun_out/fin/cfg4.err
python bench.py --no-cpu-baseline --head subpixel --batch 128 > gpurun_out/fin/cfg3_subpixel_b128.json 2> gpurun_out/fin/cfg3.err
python -c "
import json
for n in ('bench_default','cfg4_xception_os8_b16','cfg3_subpixel_b128'):
    r=json.load(open('gpurun_out/fin/%s.json'%n)); print(n, round(r['value'],1), r['roofline']['frac'], r['split_math']['value'])"
