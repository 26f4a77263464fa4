#!/bin/bash
# round 4, GPU call 11: split-K for small batches — op tests, model tests, A/B at B = 2 / 4 / 16 / 128
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4k; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "split_k or pwconv" > $out/pytest_ops.log 2>&1; echo "ops rc $?"; tail -2 $out/pytest_ops.log; grep -h "^E  " $out/pytest_ops.log | cut -c1-220 | head -12
timeout 1200 python -m pytest tests/test_gpu_model.py -q > $out/pytest_model.log 2>&1; echo "model rc $?"; tail -3 $out/pytest_model.log; grep -h "^E  \|^FAILED" $out/pytest_model.log | cut -c1-220 | head -12
bash tools/r4/ab.sh r4k/ab2 "--steps 100 --warmup 3 --batch 2" "1_sk0|DL3_SPLITK=0" "2_sk1|DL3_DY_MAT=1" "3_sk0_again|DL3_SPLITK=0" "4_sk1_again|DL3_DY_MAT=1" | tee $out/ab2.txt
bash tools/r4/ab.sh r4k/ab4 "--steps 80 --warmup 3 --batch 4" "1_sk0|DL3_SPLITK=0" "2_sk1|DL3_DY_MAT=1" | tee $out/ab4.txt
bash tools/r4/ab.sh r4k/ab16 "--steps 40 --warmup 3 --batch 16" "1_sk0|DL3_SPLITK=0" "2_sk1|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4k/ab128 "--steps 15 --warmup 3" "1_sk0|DL3_SPLITK=0" "2_sk1|DL3_DY_MAT=1" | tee $out/ab128.txt
