#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/c16; mkdir -p $out; cd $REPO
timeout 2400 python -m pytest tests -m gpu -q -x > $out/pytest_full.log 2>&1; echo "rc $?" >> $out/pytest_full.log; tail -5 $out/pytest_full.log
bash tools/ab.sh c16/b128 "--steps 20 --warmup 3 --batch 128" "1_base|DL3_LIBPATH=$REPO/build_variants/libdl3_base.so" "2_new|X=1" "3_base|DL3_LIBPATH=$REPO/build_variants/libdl3_base.so" "4_new|X=1"
