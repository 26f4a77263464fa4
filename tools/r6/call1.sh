#!/bin/bash
# round 6, call 1: op tests of the re-cut weight-gradient kernel and the exact-width last column tile, the new host tests,
# then same-call A/B of libdl3.so: build_variants/libdl3_base.so (round 5 HEAD) against the tree
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c1; mkdir -p $out
cd $REPO
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x > $out/ops.log 2>&1; echo "ops rc $?" >> $out/ops.log
tail -5 $out/ops.log
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_parallel.py -q -x \
  -k "single_image_batch or recompile or device_feed or one_rank or no_host_round or compile_accepts or batch_feeder or poison" > $out/model.log 2>&1
echo "model rc $?" >> $out/model.log
tail -5 $out/model.log
BASE=$REPO/build_variants/libdl3_base.so
bash tools/ab.sh c1/b128 "--steps 20 --warmup 3 --batch 128" "1_base|DL3_LIBPATH=$BASE" "2_new|X=1" "3_base|DL3_LIBPATH=$BASE" "4_new|X=1" "5_new_nojv|DL3_GEMM_JV=0"
bash tools/ab.sh c1/b16 "--steps 40 --warmup 5 --batch 16" "1_base|DL3_LIBPATH=$BASE" "2_new|X=1" "3_base|DL3_LIBPATH=$BASE" "4_new|X=1"
bash tools/ab.sh c1/x16 "--steps 10 --warmup 3 --batch 16 --backbone xception --os 8" "1_base|DL3_LIBPATH=$BASE" "2_new|X=1" "3_new_nojv|DL3_GEMM_JV=0" "4_base|DL3_LIBPATH=$BASE" "5_new|X=1"
