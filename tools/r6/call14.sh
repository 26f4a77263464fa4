#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 600 python -m pytest tests/gpu_ws2_probe.py -q -x 2>&1 | tail -3
bash tools/ab.sh c14/b128 "--steps 20 --warmup 3 --batch 128" "1_ws2off|DL3_WS2=0" "2_ws2on|X=1" "3_ws2off|DL3_WS2=0" "4_ws2on|X=1"
bash tools/ab.sh c14/b64 "--steps 20 --warmup 3 --batch 64" "1_ws2off|DL3_WS2=0" "2_ws2on|X=1" "3_ws2off|DL3_WS2=0" "4_ws2on|X=1"
bash tools/ab.sh c14/b32 "--steps 30 --warmup 3 --batch 32" "1_ws2off|DL3_WS2=0" "2_ws2on|X=1" "3_ws2off|DL3_WS2=0" "4_ws2on|X=1"
