# round-2 judged profiles at the bench default (B=128) + the bench lines BASELINE.md quotes + the whole GPU suite
mkdir -p gpurun_out/r2z
bash tools/collect_profiles.sh r02 128 > gpurun_out/r2z/collect.log 2>&1
python tools/pmc_family.py gpurun_out/prof_r02_pmc_FETCH_SIZE gpurun_out/prof_r02_pmc_WRITE_SIZE gpurun_out/prof_r02_plan.json > gpurun_out/r2z/r02_dw_dilated_b128_pmc.json 2> gpurun_out/r2z/pmc_family.err
python tools/pmc_family.py gpurun_out/prof_r02_pmc_FETCH_SIZE gpurun_out/prof_r02_pmc_WRITE_SIZE gpurun_out/prof_r02_plan.json gemm > gpurun_out/r2z/r02_gemm_b128_pmc.json 2>> gpurun_out/r2z/pmc_family.err
head -12 gpurun_out/r2z/r02_dw_dilated_b128_pmc.json
bash tools/collect_sq_pmc.sh 128 > gpurun_out/r2z/sq_pmc.txt 2>&1
head -6 gpurun_out/r2z/sq_pmc.txt
run() { local name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/r2z/$name.json 2> gpurun_out/r2z/$name.err; echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2z/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r.get('roofline',{}).get('frac'), r.get('roofline_hbm',{}).get('frac'), (r.get('split_math') or {}).get('value'))" 2>&1 | tail -1)"; }
run cfg3_subpixel_b128 --head subpixel --batch 128
run cfg3_original_b128 --head original --batch 128 --no-split-leg
run cfg4_xception_os8_b16 --backbone xception --os 8 --batch 16 --steps 8 --warmup 3
run cfg4_xception_os8_b32 --backbone xception --os 8 --batch 32 --steps 6 --warmup 3
run xception_os16_b16 --backbone xception --os 16 --batch 16 --steps 8 --warmup 3 --no-split-leg
for B in 2 4 8 16 32 64; do run cfg2_b$B --batch $B --no-roofline --no-split-leg --steps 30 --warmup 5; done
python bench.py --plan-json gpurun_out/r2z/plan_default.json > gpurun_out/r2z/bench_default.json 2> gpurun_out/r2z/bench_default.err
python -c "import json;r=json.load(open('gpurun_out/r2z/bench_default.json'));print('default', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline_hbm']['frac'], r['roofline_hbm']['traffic'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'], r['split_math']['value'])"
python -m pytest tests -m gpu -q -rf --durations=6 > gpurun_out/r2z/pytest_all.log 2>&1; tail -12 gpurun_out/r2z/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
