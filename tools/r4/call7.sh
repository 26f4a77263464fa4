#!/bin/bash
# round 4, GPU call 7: fine-tuning (frozen backbone) test + timing, small-batch knob sweeps
REPO=${GRAFT_REPO_ROOT:-/root/repo}
out=$REPO/gpurun_out/r4g; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "fine_tuning or mnv2_train_step or poison or frozen_bn or xception_train" > $out/pytest_model.log 2>&1; echo "model rc $?"
grep -h "passed\|failed\|^losses\|update rel-L2\|^E  " $out/pytest_model.log | cut -c1-220 | head -30
bash tools/r4/ab.sh r4g/ab16 "--steps 40 --warmup 3 --batch 16" \
  "1_default|DL3_DY_MAT=1" "2_rows65536|DL3_FUSED_ROWS=65536" "3_rows32768|DL3_FUSED_ROWS=32768" "4_wgs1024|DL3_FUSED_WGS=1024" \
  "5_dyall|DL3_DY_MAT_K=0" "6_default_again|DL3_DY_MAT=1" | tee $out/ab16.txt
bash tools/r4/ab.sh r4g/ab2 "--steps 100 --warmup 3 --batch 2" \
  "1_default|DL3_DY_MAT=1" "2_rows32768|DL3_FUSED_ROWS=32768" "3_rows8192|DL3_FUSED_ROWS=8192" "4_dyall|DL3_DY_MAT_K=0" \
  "5_default_again|DL3_DY_MAT=1" | tee $out/ab2.txt
python - <<'PY' 2>&1 | tee gpurun_out/r4g/finetune_timing.txt
# step time of the notebook's fine-tuning configuration (backbone frozen) against full training, B = 16, 512x512
import time, numpy as np, torch
import dl3_amd
from dl3_amd import graph as G
from dl3_amd.deeplabv3p import Deeplabv3
for freeze in (False, True):
    G.clear_session(seed=1)
    m = Deeplabv3(weights=None, input_shape=(512, 512, 3), classes=21, backbone="mobilenetv2", OS=16)
    if freeze:
        flag = 0
        for l in m.layers:
            l.trainable = False
            if l.name == "concat_projection":
                flag = 1
            if flag:
                l.trainable = True
    B = 16
    eng = m._engine(B, True, dropout=True)
    rng = np.random.default_rng(0)
    eng.set_input(rng.integers(0, 256, (B, 512, 512, 3)).astype(np.float32))
    y = rng.integers(0, 22, (B, 512 * 512)).astype(np.float32)
    eng.set_targets(y, (y < 21).astype(np.float32))
    for _ in range(4):
        eng.fwd_bwd(); eng.adam(None, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        eng.fwd_bwd(); eng.adam(None, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print("B=16 512x512 mnv2 %s: %.2f ms/step = %.0f img/s, %d backward launches" % (
        "notebook fine-tuning (frozen up to concat_projection)" if freeze else "everything trainable", 1e3 * dt, B / dt, len(eng.ops_bwd)))
PY
