#!/bin/bash
# round 4, GPU call 8: BatchNorm-backward sums of Xception's `sum` shortcuts inside the depthwise backward launch
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4h; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "dwconv" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -2 $out/pytest_ops.log; grep -h "^E  " $out/pytest_ops.log | cut -c1-200 | head
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "xception or fine_tuning or poison" > $out/pytest_model.log 2>&1; echo "model rc $?"; tail -2 $out/pytest_model.log; grep -h "^E  " $out/pytest_model.log | cut -c1-200 | head
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg4_xception_os8_256_train_step and not split" > $out/pytest_full.log 2>&1; echo "fullsize rc $?"
grep -h "passed\|failed\|rel err\|rel-L2 whole\|^E  " $out/pytest_full.log | cut -c1-220 | head
bash tools/r4/ab.sh r4h/abx "--steps 6 --warmup 3 --batch 16 --backbone xception --os 8" \
  "1_alias0|DL3_DW_ALIAS=0" "2_alias1|DL3_DW_ALIAS=1" "3_alias0_again|DL3_DW_ALIAS=0" "4_alias1_again|DL3_DW_ALIAS=1" | tee $out/abx.txt
