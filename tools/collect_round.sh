#!/bin/bash
# a round's judged artefacts (run through gpurun from the repo root): the -m gpu suite, bench lines, rocprofv3 kernel
# stats, PMC passes.  usage: bash tools/collect_round.sh [OUTDIR under gpurun_out, default final]; then, back in the
# container, python tools/publish_profiles.py <round number> [OUTDIR] copies the summaries into profiles/rNN_*.
set -u
REPO=$(pwd); out=$REPO/gpurun_out/${1:-final}; mkdir -p $out
export TMPDIR=/tmp
cd $REPO
# 0. the whole -m gpu suite, as the driver runs it
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -3 $out/pytest_gpu.log
# 1. the driver's own command (all legs)
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" >> $out/smoke.log; tail -2 $out/smoke.log
# 2. bench lines of the other configurations (no CPU legs)
L=$out/bench_lines.jsonl; : > $L
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
for b in 2 4 8 16 32 64; do $B --no-split-leg --no-roofline --no-legs --batch $b 2>/dev/null | tail -1 >> $L; done
$B --head subpixel --plan-json $out/plan_cfg3_subpixel.json 2>/dev/null | tail -1 >> $L
$B --head original --no-split-leg 2>/dev/null | tail -1 >> $L
$B --backbone xception --os 8 --batch 16 --steps 10 --plan-json $out/plan_cfg4_b16.json 2>/dev/null | tail -1 >> $L
$B --backbone xception --os 8 --batch 32 --steps 10 2>/dev/null | tail -1 >> $L
$B --backbone xception --os 16 --batch 16 --steps 10 --no-split-leg --no-roofline 2>/dev/null | tail -1 >> $L
# 3. rocprofv3 kernel stats of the same commands
cd /tmp
prof() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o bench -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --steps 10 --warmup 3 --plan-json $out/plan_$tag.json "$@" > $out/prof_$tag.log 2>&1; }
prof cfg2_b128
prof cfg2_b16 --batch 16
prof cfg2_b2 --batch 2 --steps 20
prof cfg3_subpixel_b128 --head subpixel
prof cfg4_b16 --backbone xception --os 8 --batch 16 --steps 6
DL3_GEMM_MATH=split prof cfg2_split_b128
# 4. PMC passes (own runs, --kernel-trace only)
pmc() { tag=$1; ctr=$2; shift 2; timeout 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${tag}_$(echo $ctr | tr ' ' '_') -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --no-roofline --no-graph --steps 3 --warmup 2 "$@" > $out/pmc_${tag}.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do pmc cfg2_b128 $c; pmc cfg4_b16 $c --backbone xception --os 8 --batch 16; done
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_cfg2_b128_SQ -o sq -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --no-roofline --no-graph --steps 3 --warmup 2 > $out/pmc_sq.log 2>&1
# 4b. per-launch PMC of the HBM-bound 1x1-convolution shapes (cold operands, tools/r5/pw_hbm_bench.py; REPS=3 + 2 warm-ups)
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  REPS=3 timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pwpmc_$i -o pmc -- python $REPO/tools/r5/pw_hbm_bench.py > $out/pwpmc_$i.log 2>&1
done
python $REPO/tools/r5/pw_hbm_bench.py > $out/pw_hbm_bench.txt 2>&1
cd $REPO
python tools/r5/pw_hbm_pmc.py $out/pwpmc_1.log $out/pwpmc_1 $out/pwpmc_2 $out/pwpmc_3 $out/pwpmc_4 $out/pwpmc_5 > $out/pw_hbm_pmc.txt 2> $out/pw_hbm_pmc.err
# 5. summaries
for t in cfg2_b128 cfg4_b16; do
  python tools/pmc_family.py $out/pmc_${t}_FETCH_SIZE $out/pmc_${t}_WRITE_SIZE $out/plan_$t.json gemm > $out/${t}_gemm_pmc.json 2> $out/${t}_gemm_pmc.err
  python tools/pmc_family.py $out/pmc_${t}_FETCH_SIZE $out/pmc_${t}_WRITE_SIZE $out/plan_$t.json > $out/${t}_dw_dilated_pmc.json 2> $out/${t}_dw_pmc.err
done
python tools/pmc_kernels.py $out/pmc_cfg2_b128_SQ 18 > $out/cfg2_b128_sq_pmc.txt 2>&1
for t in cfg2_b128 cfg2_b16 cfg2_b2 cfg3_subpixel_b128 cfg4_b16 cfg2_split_b128; do f=$(find $out/prof_$t -name "*kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; tail -1 $out/prof_$t.log | cut -c1-200; done
# raw PMC csv dirs are large: keep only the summaries
rm -rf $out/pmc_cfg* $out/prof_*/ $out/pwpmc_?/ 2>/dev/null
ls -la $out | head -40
