#!/bin/bash
# round 4, GPU call 2: re-run of the tests that failed in call 1, frozen-mode diagnostics, dY rule A/B (B=128/16/2, cfg4)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4b; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "h5 or fork or three_train or frozen or poison" > $out/pytest_model.log 2>&1; echo "model rc $?"
grep -h "passed\|failed\|^losses\|update distance" $out/pytest_model.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "pwconv or dwconv or xent" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -3 $out/pytest_ops.log
for c in 1 0; do DL3_BN_CENTER=$c timeout 600 python tools/r4/frozen_diag.py frozen > $out/diag_frozen_center$c.txt 2>&1; cat $out/diag_frozen_center$c.txt | head -12; done
timeout 600 python tools/r4/frozen_diag.py batch > $out/diag_batch.txt 2>&1; head -8 $out/diag_batch.txt
NF=$REPO/build_variants/libdl3_nofence.so
bash tools/r4/ab.sh r4b/ab128 "--steps 15 --warmup 3" \
  "1_dy0|DL3_DY_MAT=0" "2_rule|DL3_DY_MAT=1" "3_all|DL3_DY_MAT_K=0" "4_narrow|DL3_DY_MAT_K=100000" "5_k160|DL3_DY_MAT_K=160" \
  "6_rule_again|DL3_DY_MAT=1" | tee $out/ab128.txt
bash tools/r4/ab.sh r4b/ab16 "--steps 40 --warmup 3 --batch 16" \
  "1_dy0|DL3_DY_MAT=0" "2_rule|DL3_DY_MAT=1" "3_all|DL3_DY_MAT_K=0" "4_nofence_rule|DL3_LIBPATH=$NF" | tee $out/ab16.txt
bash tools/r4/ab.sh r4b/ab2 "--steps 100 --warmup 3 --batch 2" \
  "1_dy0|DL3_DY_MAT=0" "2_rule|DL3_DY_MAT=1" "3_all|DL3_DY_MAT_K=0" "4_nofence_rule|DL3_LIBPATH=$NF" | tee $out/ab2.txt
bash tools/r4/ab.sh r4b/abx "--steps 6 --warmup 3 --batch 16 --backbone xception --os 8" \
  "1_dy0|DL3_DY_MAT=0" "2_rule|DL3_DY_MAT=1" "3_all|DL3_DY_MAT_K=0" | tee $out/abx.txt
