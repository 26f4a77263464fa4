#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c20; mkdir -p $out; cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "weight_stationary_short" 2>&1 | tail -4
S="bwd2:524288x64x384 bwd2:262144x64x384"
for rep in 1 2; do
for v in "DL3_WS2=0" "DL3_WS2=1"; do echo "## $v"; env $v python tools/r6/gemm_bench.py $S; done
done 2>&1 | grep -v amdgpu.ids | tee $out/ws2_two.txt
bash tools/ab.sh c20/b128 "--steps 20 --warmup 3 --batch 128" "1_base|DL3_LIBPATH=$REPO/build_variants/libdl3_base.so" "2_new|X=1" "3_base|DL3_LIBPATH=$REPO/build_variants/libdl3_base.so" "4_new|X=1"
