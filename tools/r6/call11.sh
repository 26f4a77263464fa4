#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c11; mkdir -p $out; cd $REPO
DL3_WS2=1 timeout 600 python -m pytest tests/gpu_ws2_probe.py -q -x 2>&1 | tail -3
DL3_LIBPATH=$REPO/build_variants/libdl3_nw8.so DL3_WS2=1 timeout 600 python -m pytest tests/gpu_ws2_probe.py -q -x 2>&1 | tail -3
S="fwd:524288x160x960 fwd:524288x96x576 fwd:262144x160x960"
for rep in 1 2; do
echo "## stream"; DL3_WS2=0 python tools/r6/gemm_bench.py $S
echo "## ws2 NW=12 direct epilogue"; DL3_WS2=1 python tools/r6/gemm_bench.py $S
echo "## ws2 NW=8 direct epilogue"; DL3_LIBPATH=$REPO/build_variants/libdl3_nw8.so DL3_WS2=1 python tools/r6/gemm_bench.py $S
done 2>&1 | grep -v amdgpu.ids | tee $out/ws2.txt
