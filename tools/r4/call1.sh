#!/bin/bash
# round 4, GPU call 1: the whole -m gpu suite, the default bench line, the dY / epilogue A/B at B=128, the centred
# frozen BatchNorm A/B on the Xception 512x512 inference test
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4a; mkdir -p $out
cd $REPO
timeout 1500 python -m pytest tests -m gpu -q --maxfail=8 --durations=15 > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
tail -c 600 $out/bench_default.json
NF=$REPO/build_variants/libdl3_nofence.so
bash tools/r4/ab.sh r4a/ab "--steps 15 --warmup 3" \
  "1_nofence_dy0|DL3_LIBPATH=$NF DL3_DY_MAT=0" \
  "2_fence_dy0|DL3_DY_MAT=0" \
  "3_dy320|DL3_DY_MAT=320" \
  "4_dyall|DL3_DY_MAT=100000" \
  "5_dyall_epi3off|DL3_DY_MAT=100000 DL3_GEMM_EPI3=0" \
  "6_dyall_preoff|DL3_DY_MAT=100000 DL3_GEMM_PRE=0" \
  "7_dy320_preoff|DL3_DY_MAT=320 DL3_GEMM_PRE=0" | tee $out/ab_summary.txt
DL3_BN_CENTER=0 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg4_xception_os8_512_forward" > $out/center0.log 2>&1
grep -h "flips\|logits rel" $out/center0.log | sed 's/^/center=0: /'
DL3_BN_CENTER=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg4_xception_os8_512_forward" > $out/center1.log 2>&1
grep -h "flips\|logits rel" $out/center1.log | sed 's/^/center=1: /'
