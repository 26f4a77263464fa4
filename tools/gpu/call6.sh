# round-2 judged profiles + the bench lines BASELINE.md quotes
mkdir -p gpurun_out/r2f
bash tools/collect_profiles.sh r02 64 > gpurun_out/r2f/collect.log 2>&1
python tools/pmc_family.py gpurun_out/prof_r02_pmc_FETCH_SIZE gpurun_out/prof_r02_pmc_WRITE_SIZE gpurun_out/prof_r02_plan.json > gpurun_out/r2f/r02_dw_dilated_b64_pmc.json 2> gpurun_out/r2f/pmc_family.err
cat gpurun_out/r2f/r02_dw_dilated_b64_pmc.json | head -30
bash tools/collect_sq_pmc.sh 64 > gpurun_out/r2f/sq_pmc.txt 2>&1
head -12 gpurun_out/r2f/sq_pmc.txt
run() { local name=$1; shift; python bench.py --no-cpu-baseline "$@" > gpurun_out/r2f/$name.json 2> gpurun_out/r2f/$name.err; echo "$name: $(python -c "import json;r=json.load(open('gpurun_out/r2f/$name.json'));print(round(r['value'],1), round(r['ms_per_step'],3), r.get('roofline',{}).get('frac'), r.get('roofline_hbm',{}).get('frac'))" 2>&1 | tail -1)"; }
run cfg3_subpixel_b64 --head subpixel --batch 64
run cfg3_original_b64 --head original --batch 64
run cfg4_xception_os8_b16 --backbone xception --os 8 --batch 16 --steps 8 --warmup 3
run cfg4_xception_os16_b16 --backbone xception --os 16 --batch 16 --steps 8 --warmup 3
for B in 2 4 8 16 32 128; do run cfg2_b$B --batch $B --no-roofline --steps 30 --warmup 5; done
python bench.py --plan-json gpurun_out/r2f/plan_default.json > gpurun_out/r2f/bench_default.json 2> gpurun_out/r2f/bench_default.err
tail -c 900 gpurun_out/r2f/bench_default.json
