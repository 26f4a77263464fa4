#!/bin/bash
# Texture-addresser (vector-memory address path) counters of the whole training step per kernel: is a GEMM kernel limited
# by the number of cache lines its wave-loads touch rather than by HBM or the matrix pipe?  Own passes, --kernel-trace only.
# usage (through gpurun, from the repo root): bash tools/collect_ta_pmc.sh [batch]
B=${1:-64}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > $REPO/gpurun_out/list_avail.txt 2>&1
pass() {
  local tag=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/prof_ta_$tag -o ta -- python $REPO/bench.py --batch $B --no-cpu-baseline --no-split-leg --no-roofline --no-graph --steps 2 --warmup 2 > $REPO/gpurun_out/prof_ta_$tag.log 2>&1
  cd $REPO; python tools/pmc_kernels.py gpurun_out/prof_ta_$tag 14; cd /tmp
}
pass busy TA_TA_BUSY_sum TA_BUSY_avr
pass stall TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
