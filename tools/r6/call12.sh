#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
DL3_WS2=1 timeout 600 python -m pytest tests/gpu_ws2_probe.py -q -x 2>&1 | tail -3
bash tools/ab.sh c12/b128 "--steps 20 --warmup 3 --batch 128" "1_ws2off|DL3_WS2=0" "2_ws2on|DL3_WS2=1" "3_ws2off|DL3_WS2=0" "4_ws2on|DL3_WS2=1"
bash tools/ab.sh c12/x16 "--steps 10 --warmup 3 --batch 16 --backbone xception --os 8" "1_new|X=1" "2_base|DL3_LIBPATH=$REPO/build_variants/libdl3_base.so" "3_new|X=1"
