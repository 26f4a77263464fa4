"""Round 6: the Xception OS=8 512x512 argmax question with statistics — N single-image forwards (inference
BatchNorm, moving statistics calibrated once on a 2-image batch as tests/test_gpu_fullsize.py does), argmax flips of the HIP
path and of the torch oracle's fp32 run against the float64 oracle, per image and summed, with a sign test.

  python tools/r6/xception_argmax_stats.py [N=16]"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_num_threads(min(32, os.cpu_count() or 1))
import dl3_amd  # noqa: E402,F401
from dl3_amd import graph as G  # noqa: E402
from dl3_amd.deeplabv3p import Deeplabv3  # noqa: E402
from oracle import dl3_oracle as O  # noqa: E402
from oracle import torch_ref as T  # noqa: E402
from tests.test_gpu_fullsize import _data  # noqa: E402
from tests.test_gpu_model import _load  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape, classes = (512, 512, 3), 21
model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone="xception", OS=8)
kw = dict(backbone="xception", input_shape=shape, classes=classes, OS=8)
params = O.init_params(O.param_shapes("xception", classes), seed=1)
x, _, _ = _data(shape, 2, classes, seed=2)
params = T.calibrate_bn(params, x, dtype=torch.float32, **kw)
_load(model, params)
rows = []
for i in range(N):
    x1 = x[:1] if i == 0 else _data(shape, 1, classes, seed=2 + i)[0]
    model.predict(x1, batch_size=1)
    got = model._active.logits()
    mask = model._active.argmax()
    ref = T.infer_logits(params, x1, dtype=torch.float64, **kw)
    ref32 = T.infer_logits(params, x1, dtype=torch.float32, **kw)
    want = ref.argmax(-1)
    fg, f32 = int((mask != want).sum()), int((ref32.argmax(-1) != want).sum())
    eg = float(np.abs(got - ref).max() / np.abs(ref).max())
    e32 = float(np.abs(ref32 - ref).max() / np.abs(ref).max())
    rows.append((fg, f32, eg, e32))
    print("image %2d: argmax flips gpu %3d / torch-fp32 %3d of %d; logits rel err gpu %.2e / torch-fp32 %.2e" % (
        i, fg, f32, want.size, eg, e32))
    sys.stdout.flush()
sg, s32 = sum(r[0] for r in rows), sum(r[1] for r in rows)
wins = sum(r[0] < r[1] for r in rows)
losses = sum(r[0] > r[1] for r in rows)
n = wins + losses
# two-sided sign test: P(at least this lopsided | fair coin)
k = max(wins, losses)
p = min(1.0, 2.0 * sum(math.comb(n, j) for j in range(k, n + 1)) / 2 ** n) if n else 1.0
print("sum over %d images: gpu %d, torch-fp32 %d (ratio %.3f); gpu better on %d, worse on %d, tied %d: sign test p = %.3f" % (
    N, sg, s32, sg / max(s32, 1), wins, losses, N - n, p))
print("mean logits rel err: gpu %.3e, torch-fp32 %.3e" % (np.mean([r[2] for r in rows]), np.mean([r[3] for r in rows])))
