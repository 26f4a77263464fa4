#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash tools/ab.sh c6/b16 "--steps 40 --warmup 5 --batch 16" "1_default|DL3_GEMM_JV=0" "2_py1024|DL3_GEMM_JV=0 DL3_GEMM_PY=1024" "3_py2048|DL3_GEMM_JV=0 DL3_GEMM_PY=2048" "4_py4096|DL3_GEMM_JV=0 DL3_GEMM_PY=4096" "5_default|DL3_GEMM_JV=0"
bash tools/ab.sh c6/x16 "--steps 10 --warmup 3 --batch 16 --backbone xception --os 8" "1_default|DL3_GEMM_JV=0" "2_py1024|DL3_GEMM_JV=0 DL3_GEMM_PY=1024" "3_py2560|DL3_GEMM_JV=0 DL3_GEMM_PY=2560" "4_py4096|DL3_GEMM_JV=0 DL3_GEMM_PY=4096" "5_default|DL3_GEMM_JV=0"
bash tools/ab.sh c6/b128 "--steps 20 --warmup 3 --batch 128" "1_default|DL3_GEMM_JV=0" "2_py1024|DL3_GEMM_JV=0 DL3_GEMM_PY=1024" "3_py4096|DL3_GEMM_JV=0 DL3_GEMM_PY=4096" "4_py16384|DL3_GEMM_JV=0 DL3_GEMM_PY=16384"
