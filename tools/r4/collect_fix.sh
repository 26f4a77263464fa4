#!/bin/bash
# round 4: the cfg2 kernel-stats and PMC passes again, without the compact legs in the profiled process (their launches
# polluted the per-kernel table, and HIP events between launches under --pmc abort the queue: "AQL packet is malformed")
set -u
REPO=$(pwd); out=$REPO/gpurun_out/r4final; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
prof() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o bench -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --steps 10 --warmup 3 --plan-json $out/plan_$tag.json "$@" > $out/prof_$tag.log 2>&1; }
prof cfg2_b128
DL3_GEMM_MATH=split prof cfg2_split_b128
pmc() { tag=$1; ctr=$2; shift 2; timeout 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/pmc_${tag}_$(echo $ctr | tr ' ' '_') -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --no-roofline --no-graph --steps 3 --warmup 2 "$@" > $out/pmc_${tag}_$ctr.log 2>&1; }
for c in FETCH_SIZE WRITE_SIZE; do pmc cfg2_b128 $c; done
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_cfg2_b128_SQ -o sq -- python $REPO/bench.py --no-cpu-baseline --no-split-leg --no-legs --no-roofline --no-graph --steps 3 --warmup 2 > $out/pmc_sq.log 2>&1
cd $REPO
t=cfg2_b128
python tools/pmc_family.py $out/pmc_${t}_FETCH_SIZE $out/pmc_${t}_WRITE_SIZE $out/plan_$t.json gemm > $out/${t}_gemm_pmc.json 2> $out/${t}_gemm_pmc.err
python tools/pmc_family.py $out/pmc_${t}_FETCH_SIZE $out/pmc_${t}_WRITE_SIZE $out/plan_$t.json > $out/${t}_dw_dilated_pmc.json 2> $out/${t}_dw_pmc.err
python tools/pmc_kernels.py $out/pmc_cfg2_b128_SQ 20 > $out/cfg2_b128_sq_pmc.txt 2>&1
for t in cfg2_b128 cfg2_split_b128; do f=$(find $out/prof_$t -name "*kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; tail -1 $out/prof_$t.log | cut -c1-200; done
rm -rf $out/pmc_cfg* $out/prof_*/ 2>/dev/null
cat $out/cfg2_b128_gemm_pmc.json | tail -8; cat $out/cfg2_b128_dw_dilated_pmc.json | tail -8; head -12 $out/cfg2_b128_sq_pmc.txt
