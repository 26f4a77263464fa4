"""where does the HIP path's Xception OS=8 512x512 inference part from float64 more than torch-fp32 does?
Per BatchNormalization layer, in graph order: distance to the float64 oracle of (a) the GPU's tensor as its consumers read
it (scale * stored + shift) and (b) the torch-fp32 oracle's — relative L2 and max-abs over max-abs.  The layer where the
ratio (a) / (b) jumps is the one to fix.

  python tools/r5/xception_layer_distance.py [--size 512] > profiles/r05_xception_layer_distance.txt"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dl3_amd  # noqa: E402,F401
from dl3_amd import graph as G  # noqa: E402
from dl3_amd.deeplabv3p import Deeplabv3  # noqa: E402
from dl3_amd.engine import V_SCALE, V_SHIFT  # noqa: E402
from oracle import dl3_oracle as O  # noqa: E402
from oracle import torch_ref as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--backbone", default="xception")
ap.add_argument("--os", type=int, default=8)
ap.add_argument("--seed", type=int, default=2, help="seed of the input images (the BatchNorm statistics are calibrated on them)")
ap.add_argument("--brief", action="store_true", help="only the ASPP / decoder rows and the flip counts")
a = ap.parse_args()
torch.set_num_threads(min(32, os.cpu_count() or 1))
shape, classes = (a.size, a.size, 3), 21
kw = dict(backbone=a.backbone, input_shape=shape, classes=classes, OS=a.os)
G.clear_session()
model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone=a.backbone, OS=a.os)
params = O.init_params(O.param_shapes(a.backbone, classes), seed=1)
rng = np.random.default_rng(a.seed)
x = rng.integers(0, 256, (2,) + shape).astype(np.float32)
params = T.calibrate_bn(params, x, dtype=torch.float32, **kw)
for l in model.layers:
    if l.weights:
        l.set_weights([params[n] for n in l.weights])
x1 = x[:1]
model.predict(x1, batch_size=1)
eng = model._active
got = eng.logits()


def oracle(dtype):
    with torch.no_grad():
        ref = T.Ref(params, False, dtype=dtype)
        ref.record = {}
        lg = ref.logits(x1, **kw).numpy()
    return ref.record, lg


r64, l64 = oracle(torch.float64)
r32, l32 = oracle(torch.float32)


def dist(a_, b_):
    a_, b_ = np.asarray(a_, np.float64), np.asarray(b_, np.float64)
    return float(np.linalg.norm(a_ - b_) / (np.linalg.norm(b_) + 1e-30)), float(np.abs(a_ - b_).max() / (np.abs(b_).max() + 1e-30))


print("# seed %d" % a.seed)
print("# %s OS=%d %dx%d B=1 inference: distance to the float64 oracle per BatchNormalization output" % (a.backbone, a.os, a.size, a.size))
print("# %-52s %11s %11s %7s | %11s %11s" % ("layer", "gpu relL2", "fp32 relL2", "ratio", "gpu max", "fp32 max"))
order = [l.name for l in model.layers if l.kind == "BatchNormalization"]
bybn = {}
for buf in eng.bufs:
    for bn, off, C in buf.bns:
        bybn[bn.name] = (buf, off, C)
for name in order:
    if name not in bybn or name not in r64:
        continue
    buf, off, C = bybn[name]
    t = buf.t.view(buf.M, buf.ld)[:, off:off + C].double()
    v = (t * buf.vec[V_SCALE, off:off + C].double() + buf.vec[V_SHIFT, off:off + C].double()).cpu().numpy()
    ref = r64[name].reshape(-1, r64[name].shape[-1])
    c = min(C, ref.shape[1])   # (stored width may exceed the logical one: 736 vs 728)
    g2, gm = dist(v[:, :c], ref[:, :c])
    f2, fm = dist(r32[name].reshape(ref.shape)[:, :c], ref[:, :c])
    if not a.brief or name.startswith(("aspp", "image_pooling", "concat", "decoder", "feature_projection")):
        print("%-54s %11.3e %11.3e %7.2f | %11.3e %11.3e" % (name, g2, f2, g2 / max(f2, 1e-30), gm, fm))
g2, gm = dist(got, l64)
f2, fm = dist(l32, l64)
print("%-54s %11.3e %11.3e %7.2f | %11.3e %11.3e" % ("logits (full resolution)", g2, f2, g2 / max(f2, 1e-30), gm, fm))
want = l64.argmax(-1)
print("seed %d argmax flips: gpu %d, torch-fp32 %d of %d" % (a.seed, int((got.argmax(-1) != want).sum()), int((l32.argmax(-1) != want).sum()), want.size))
