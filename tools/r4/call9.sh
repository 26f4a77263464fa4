#!/bin/bash
# round 4, GPU call 9: centred frozen BatchNorm behind depthwise convolutions too (DL3_BN_CENTER=2) — does it close the
# remaining argmax-flip excess of the Xception 512x512 inference test?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4i; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "dwconv_fwd" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -2 $out/pytest_ops.log; grep -h "^E  " $out/pytest_ops.log | cut -c1-200 | head
for c in 1 2 0; do
  DL3_BN_CENTER=$c timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k "cfg4_xception_os8_512_forward" > $out/center$c.log 2>&1
  grep -h "flips\|logits rel\|passed\|failed" $out/center$c.log | sed "s/^/center=$c: /" | cut -c1-260
done
DL3_BN_CENTER=2 timeout 900 python -m pytest tests/test_gpu_model.py -q -k "cfg1 or frozen_bn or h5 or inference or odd_and or alpha" > $out/pytest_model.log 2>&1; echo "model(center=2) rc $?"; tail -2 $out/pytest_model.log
