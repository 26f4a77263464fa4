#!/bin/bash
# round 5, GPU call 9: which of the depthwise kernels gain from the straight-line interior groups (DL3_DW_FAST bit mask)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
O=gpurun_out/c9
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 200 python tools/r5/dw_bench.py > $O/dw_old.txt 2>&1
for f in 0 1 2 4 7; do DL3_DW_FAST=$f timeout 200 python tools/r5/dw_bench.py > $O/dw_fast$f.txt 2>&1; done
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 200 python tools/r5/dw_bench.py 16 > $O/dw16_old.txt 2>&1
for f in 0 7; do DL3_DW_FAST=$f timeout 200 python tools/r5/dw_bench.py 16 > $O/dw16_fast$f.txt 2>&1; done
for f in old fast0 fast1 fast2 fast4 fast7; do echo "== $f"; grep "^dw" $O/dw_$f.txt | cut -c1-60,84-112; done
for f in old fast0 fast7; do echo "== B=16 $f"; grep "^dw" $O/dw16_$f.txt | cut -c1-60,84-112; done
