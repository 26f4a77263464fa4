#!/bin/bash
# round 3, GPU call 1: new parity tests + backward-fork experiment matrix (bench lines only, no CPU legs)
out=gpurun_out/r3a; mkdir -p $out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $out/dev.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "subpixel_with_kernel or compile_accepts or train_step_gradients" -s > $out/t_model.log 2>&1; echo "rc $?" >> $out/t_model.log
timeout 900 python -m pytest tests/test_gpu_parallel.py -q -x -m gpu -s > $out/t_par.log 2>&1; echo "rc $?" >> $out/t_par.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-leg --no-roofline"
run() { name=$1; shift; env "$@" $B > $out/b_$name.json 2> $out/b_$name.err; tail -c 600 $out/b_$name.json | head -c 300; echo; }
run nofork DL3_FORK=0
run fork DL3_FORK=1
run fork_w256 DL3_FORK=1 DL3_WGRAD_WGS=256
run fork_w512 DL3_FORK=1 DL3_WGRAD_WGS=512
run fork_w256_g256 DL3_FORK=1 DL3_WGRAD_WGS=256 DL3_BWD_GEMM_PY=256
run fork_w512_g256 DL3_FORK=1 DL3_WGRAD_WGS=512 DL3_BWD_GEMM_PY=256
for f in $out/b_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(r["value"],1), round(r["ms_per_step"],2), r["config"].get("backward_fork"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done > $out/summary.txt
cat $out/summary.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -s -k "cfg4_xception_os8_512_train or original" > $out/t_full.log 2>&1; echo "rc $?" >> $out/t_full.log
tail -30 $out/t_full.log
