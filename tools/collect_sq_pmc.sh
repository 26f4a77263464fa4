#!/bin/bash
# SQ counters of the whole training step per kernel (own pass, --kernel-trace only): matrix-pipe busy cycles, wave cycles,
# instruction-wait cycles.  usage (through gpurun, from the repo root): bash tools/collect_sq_pmc.sh [batch]
B=${1:-64}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof_sq -o sq -- python $REPO/bench.py --batch $B --no-cpu-baseline --no-split-leg --no-roofline --no-graph --steps 3 --warmup 2 > $REPO/gpurun_out/prof_sq.log 2>&1
cd $REPO
python tools/pmc_kernels.py gpurun_out/prof_sq 16
