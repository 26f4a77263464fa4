mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q -rf -k "not parallel and not mnv2_512" > gpurun_out/r2h/pytest.log 2>&1; tail -6 gpurun_out/r2h/pytest.log
for P in 1 0; do
  DL3_CHANNEL_PAD=$P python bench.py --backbone xception --os 8 --batch 16 --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/r2h/cfg4_pad$P.json 2> gpurun_out/r2h/cfg4_pad$P.err
  python -c "
import json;r=json.load(open('gpurun_out/r2h/cfg4_pad$P.json'));print('cfg4 pad=$P', round(r['value'],2), round(r['ms_per_step'],2), 'mfma', round(r['roofline']['frac'],3), 'hbm', round(r['roofline_hbm']['frac'],3), r['insitu_share'])"
done
DL3_CHANNEL_PAD=1 python bench.py --backbone xception --os 16 --batch 16 --no-cpu-baseline --no-roofline --steps 8 --warmup 3 | cut -c1-160
