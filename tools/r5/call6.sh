#!/bin/bash
# round 5, GPU call 6: the explicit few-row entry point (dl3_pwconv_fwd_rows) in the model tests; Xception forward distance
# to float64 on three more inputs (is the argmax gap to torch-fp32 systematic or the draw of one image?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
O=gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "few_rows or pwconv_fwd" > $O/pytest_ops.log 2>&1
echo "ops rc=$?" > $O/status.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -s > $O/pytest_model.log 2>&1
echo "model rc=$?" >> $O/status.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "cfg4_xception_os8_512_forward" > $O/pytest_full.log 2>&1
echo "full rc=$?" >> $O/status.txt
for s in 3 4 5; do
timeout 600 python tools/r5/xception_layer_distance.py --seed $s --brief > $O/xception_seed$s.txt 2> $O/xception_seed$s.err
done
cat $O/status.txt
for f in $O/pytest_ops.log $O/pytest_full.log; do echo "== $f"; tail -n 6 $f; done
echo "== model"; grep -n "frozen-BN\|passed\|failed\|FAILED\|Error" $O/pytest_model.log | head -20
for s in 3 4 5; do echo "== seed $s"; tail -n 9 $O/xception_seed$s.txt; tail -n 2 $O/xception_seed$s.err; done
