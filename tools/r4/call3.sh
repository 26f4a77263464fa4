#!/bin/bash
# round 4, GPU call 3: the fused backward kernel — op tests, model tests, A/B at B = 128 / 16 / 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4c; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "fused or pwconv" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -4 $out/pytest_ops.log
grep -h "^E  " $out/pytest_ops.log | head -20
timeout 1200 python -m pytest tests/test_gpu_model.py -q -s -k "mnv2_train_step or poison or three_train or adam or h5 or fork or xception_train" > $out/pytest_model.log 2>&1; echo "model rc $?"
grep -h "passed\|failed\|^losses\|update distance\|moments\|^E  " $out/pytest_model.log | head -40
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg2_cfg3_mnv2_512_train_step and deeplab" > $out/pytest_full.log 2>&1; echo "fullsize rc $?"
grep -h "passed\|failed\|rel err\|rel-L2 whole\|^E  " $out/pytest_full.log | head
bash tools/r4/ab.sh r4c/ab128 "--steps 15 --warmup 3" \
  "1_fused0|DL3_FUSED_BWD=0" "2_fused1|DL3_FUSED_BWD=1" "3_fused1_wgs2048|DL3_FUSED_WGS=2048" "4_fused1_wgs512|DL3_FUSED_WGS=512" \
  "5_fused0_again|DL3_FUSED_BWD=0" | tee $out/ab128.txt
bash tools/r4/ab.sh r4c/ab16 "--steps 40 --warmup 3 --batch 16" "1_fused0|DL3_FUSED_BWD=0" "2_fused1|DL3_FUSED_BWD=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4c/ab2 "--steps 100 --warmup 3 --batch 2" "1_fused0|DL3_FUSED_BWD=0" "2_fused1|DL3_FUSED_BWD=1" \
  "3_fused_all|DL3_FUSED_ROWS=1" | tee $out/ab2.txt
