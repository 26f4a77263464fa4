#!/bin/bash
# round 4, GPU call 14: operand ring (PD K-tiles in flight) in the 32-row stream kernels — op tests per build, A/B at B = 2 / 4 / 16
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4n; mkdir -p $out
cd $REPO
K="every_tile_configuration or test_pwconv_fwd or test_pwconv_bwd_data"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "$K" > $out/pytest_base.log 2>&1; echo "base rc $?"; tail -1 $out/pytest_base.log
for pd in 2 3; do
  DL3_LIBPATH=$REPO/build_variants/libdl3_pd$pd.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "$K" > $out/pytest_pd$pd.log 2>&1
  echo "pd$pd rc $?"; tail -1 $out/pytest_pd$pd.log; grep -h "^E  \|^FAILED" $out/pytest_pd$pd.log | cut -c1-200 | head -8
done
V2=$REPO/build_variants/libdl3_pd2.so; V3=$REPO/build_variants/libdl3_pd3.so
bash tools/r4/ab.sh r4n/ab2 "--steps 100 --warmup 3 --batch 2" "1_pd1|DL3_DY_MAT=1" "2_pd2|DL3_LIBPATH=$V2" "3_pd3|DL3_LIBPATH=$V3" \
   "4_pd1_again|DL3_DY_MAT=1" "5_pd2_again|DL3_LIBPATH=$V2" "6_pd3_again|DL3_LIBPATH=$V3" | tee $out/ab2.txt
bash tools/r4/ab.sh r4n/ab4 "--steps 80 --warmup 3 --batch 4" "1_pd1|DL3_DY_MAT=1" "2_pd2|DL3_LIBPATH=$V2" "3_pd3|DL3_LIBPATH=$V3" | tee $out/ab4.txt
bash tools/r4/ab.sh r4n/ab16 "--steps 40 --warmup 3 --batch 16" "1_pd1|DL3_DY_MAT=1" "2_pd2|DL3_LIBPATH=$V2" "3_pd3|DL3_LIBPATH=$V3" | tee $out/ab16.txt
