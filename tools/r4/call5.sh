#!/bin/bash
# round 4, GPU call 5: 32-deep K-tiles (whole 128-byte lines of A per K-tile) for forward / single-tensor bwd-data
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4e; mkdir -p $out
cd $REPO
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "fused" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -2 $out/pytest_ops.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "adam" > $out/pytest_model.log 2>&1; echo "model rc $?"; tail -2 $out/pytest_model.log
V=$REPO/build_variants/libdl3_kt32.so
bash tools/r4/ab.sh r4e/abx "--steps 6 --warmup 3 --batch 16 --backbone xception --os 8" \
  "1_kt16|DL3_DY_MAT=1" "2_kt32|DL3_LIBPATH=$V" "3_kt16_again|DL3_DY_MAT=1" "4_kt32_again|DL3_LIBPATH=$V" | tee $out/abx.txt
bash tools/r4/ab.sh r4e/ab128 "--steps 15 --warmup 3" \
  "1_kt16|DL3_DY_MAT=1" "2_kt32|DL3_LIBPATH=$V" "3_kt16_again|DL3_DY_MAT=1" "4_kt32_again|DL3_LIBPATH=$V" | tee $out/ab128.txt
