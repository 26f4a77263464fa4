"""GPU diagnostic (round 4): where does the frozen-BatchNorm training step differ from float64 — forward logits, every
gradient tensor, and the first Adam update — for the GPU, the torch fp32 restatement and the numpy fp32 oracle.
   python tools/r4/frozen_diag.py [batch|frozen]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dl3_oracle as O  # noqa: E402
from oracle import torch_ref as T  # noqa: E402
from tests.test_gpu_model import _build, _load  # noqa: E402


def l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "frozen"
    frozen = mode == "frozen"
    classes, B, shape = 3, 4, (64, 64, 3)
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    rng = np.random.default_rng(21)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    x0 = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    params = O.calibrate_bn(params, x0, **kw)
    _load(model, params)
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    y = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
    sw = ((y < classes) * rng.uniform(0.5, 2.0, y.shape)).astype(np.float32)
    eng = model._engine(B, True, bn_mode=mode, dropout=False, use_graph=False)
    eng.set_input(x)
    eng.set_targets(y, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    l64, g64, lg64, _ = O.train_grads(p64, x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64), bn_frozen=frozen, **kw)
    lt, gt, lgt = T.train_grads(params, x, y, sw, bn_frozen=frozen, dtype=torch.float32, **kw)
    ln, gn, lgn, _ = O.train_grads(params, x, y, sw, bn_frozen=frozen, **kw)
    print("mode %s DL3_BN_CENTER=%s DL3_DY_MAT=%s" % (mode, os.environ.get("DL3_BN_CENTER", "1"), os.environ.get("DL3_DY_MAT", "1")))
    print("logits rel-L2 to float64: gpu %.2e torch32 %.2e numpy32 %.2e | loss gpu %.8f f64 %.8f torch32 %.8f numpy32 %.8f" % (
        l2(eng.logits(), lg64), l2(lgt, lg64), l2(lgn, lg64), float(eng.loss[0].item()), l64, lt, ln))
    rows = []
    for n, g in g64.items():
        if g is None or np.abs(g).max() == 0:
            continue
        gg = eng.grad_of(n)
        rows.append((n, l2(gg, g), l2(gt[n].reshape(g.shape), g), l2(gn[n], g)))
    for kind in ("/kernel:0", "/depthwise_kernel:0", "/gamma:0", "/beta:0", "/bias:0"):
        r = [v for v in rows if v[0].endswith(kind)]
        if r:
            print("  %-20s n=%3d  median rel-L2: gpu %.2e torch32 %.2e numpy32 %.2e | max gpu %.2e (%s)" % (
                kind, len(r), np.median([v[1] for v in r]), np.median([v[2] for v in r]), np.median([v[3] for v in r]),
                max(v[1] for v in r), max(r, key=lambda v: v[1])[0]))
    order = [n for n in params if n in dict((r[0], 1) for r in rows)]
    byname = {r[0]: r for r in rows}
    print("  by depth (every 6th tensor):")
    for n in order[::6]:
        r = byname[n]
        print("    %-44s gpu %.2e torch32 %.2e numpy32 %.2e" % r)


if __name__ == "__main__":
    main()
