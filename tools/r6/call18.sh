#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c18; mkdir -p $out; cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "bwd_weight" > $out/ops.log 2>&1; echo "ops rc $?" >> $out/ops.log; tail -3 $out/ops.log
W="wgraddy:524288x160x960 wgraddy:524288x96x576 wgrad:524288x160x960 wgrad:524288x96x576 wgraddy:262144x160x960 wgraddy:65536x160x960 wgraddy:65536x96x576"
for rep in 1 2; do
for v in 0 1; do echo "## DL3_WGRAD_ROW=$v"; DL3_WGRAD_ROW=$v python tools/r6/gemm_bench.py $W; done
done 2>&1 | grep -v amdgpu.ids | tee $out/wgrad_row.txt
bash tools/ab.sh c18/b128 "--steps 20 --warmup 3 --batch 128" "1_rowoff|DL3_WGRAD_ROW=0" "2_rowon|X=1" "3_rowoff|DL3_WGRAD_ROW=0" "4_rowon|X=1"
bash tools/ab.sh c18/b16 "--steps 40 --warmup 5 --batch 16" "1_rowoff|DL3_WGRAD_ROW=0" "2_rowon|X=1" "3_rowoff|DL3_WGRAD_ROW=0" "4_rowon|X=1"
