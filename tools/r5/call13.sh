#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c13
export TMPDIR=/tmp
O=gpurun_out/c13
timeout 300 python tools/r5/odd_shape_diag.py > $O/cur.txt 2>&1
DL3_LIBPATH=$PWD/build_variants/libdl3_c6.so timeout 300 python tools/r5/odd_shape_diag.py > $O/old.txt 2>&1
DL3_LIBPATH=$PWD/build_variants/libdl3_nocontract_conv3x3.so timeout 300 python tools/r5/odd_shape_diag.py > $O/nc_conv.txt 2>&1
DL3_LIBPATH=$PWD/build_variants/libdl3_nocontract_dw.so timeout 300 python tools/r5/odd_shape_diag.py > $O/nc_dw.txt 2>&1
paste <(cut -c1-64 $O/cur.txt) <(cut -c53-64 $O/old.txt) <(cut -c53-64 $O/nc_conv.txt) <(cut -c53-64 $O/nc_dw.txt) | head -150
