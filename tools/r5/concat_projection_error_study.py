"""CPU study: the OWN rounding error of concat_projection (deeplabv3p.py:402-408; K = 1280 at Xception) under
the summation orders in play — on identical float32 inputs, against the exact float64 product of those inputs:

  torch-fp32      F.conv2d on the CPU (the yardstick of the argmax test)
  seq             one float32 accumulator per output walking k = 0..K-1 two at a time (v_mfma_f32_32x32x2_f32), the
                  image-pooling rows in double, stored as acc + (addend - mean)   [the HIP path]
  chunk C         the same with the per-pixel reduction cut into pieces of C (one GEMM launch per piece, each adding the
                  previous result: dl3_pwconv_fwd_add)

All are read the way the consumers read them: scale * stored + beta.  Inputs: the float64 oracle's ASPP branch outputs of
one random image, rounded to float32.   python tools/r5/concat_projection_error_study.py [--size 256]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dl3_oracle as O  # noqa: E402
from oracle import torch_ref as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
a = ap.parse_args()
torch.set_num_threads(min(32, os.cpu_count() or 1))
shape, classes = (a.size, a.size, 3), 21
kw = dict(backbone="xception", input_shape=shape, classes=classes, OS=8)
params = O.init_params(O.param_shapes("xception", classes), seed=1)
rng = np.random.default_rng(2)
x = rng.integers(0, 256, (2,) + shape).astype(np.float32)
params = T.calibrate_bn(params, x, dtype=torch.float32, **kw)
with torch.no_grad():
    ref = T.Ref(params, False, dtype=torch.float64)
    ref.record = {}
    ref.logits(x[:1], **kw)
rec = ref.record
relu = lambda t: np.maximum(t, 0.0)
names = ["image_pooling_BN", "aspp0_BN", "aspp1_pointwise_BN", "aspp2_pointwise_BN", "aspp3_pointwise_BN"]
pool = relu(rec[names[0]]).reshape(1, -1).astype(np.float32)                     # [1][256]
pix = np.concatenate([relu(rec[n]).reshape(-1, 256) for n in names[1:]], axis=1).astype(np.float32)   # [HW][1024]
W = params["concat_projection/kernel:0"].reshape(1280, 256).astype(np.float32)
g, b = params["concat_projection_BN/gamma:0"].astype(np.float64), params["concat_projection_BN/beta:0"].astype(np.float64)
mm, mv = params["concat_projection_BN/moving_mean:0"].astype(np.float64), params["concat_projection_BN/moving_variance:0"].astype(np.float64)
scale64 = g / np.sqrt(mv + 1e-5)
exact = (pool.astype(np.float64) @ W[:256].astype(np.float64) + pix.astype(np.float64) @ W[256:].astype(np.float64) - mm) * scale64 + b
print("# one %dx%d image, %d pixels at OS 8; |y| rms %.3f, |BN(y)| rms %.3f" % (a.size, a.size, pix.shape[0],
      np.sqrt(np.mean((pool.astype(np.float64) @ W[:256] + pix.astype(np.float64) @ W[256:]) ** 2)), np.sqrt(np.mean(exact ** 2))))


def report(label, out):
    d = np.asarray(out, np.float64) - exact
    print("%-28s rel-L2 %.3e   max-abs %.3e" % (label, np.linalg.norm(d) / np.linalg.norm(exact), np.abs(d).max()))


# torch-fp32: concat -> conv -> batch_norm, as oracle/torch_ref.py does
xin = np.concatenate([np.repeat(pool, pix.shape[0], 0), pix], axis=1)             # [HW][1280]
tx = torch.from_numpy(xin.T.reshape(1, 1280, -1, 1).copy())
tw = torch.from_numpy(W.T.reshape(256, 1280, 1, 1).copy())
ty = F.conv2d(tx, tw)
tb = F.batch_norm(ty, torch.from_numpy(mm.astype(np.float32)), torch.from_numpy(mv.astype(np.float32)),
                  torch.from_numpy(g.astype(np.float32)), torch.from_numpy(b.astype(np.float32)), False, 0.0, 1e-5)
report("torch-fp32", tb.reshape(256, -1).T.numpy())

scale32, beta32 = scale64.astype(np.float32), b.astype(np.float32)
addend = (pool.astype(np.float64) @ W[:256].astype(np.float64) - mm).astype(np.float32)     # per-image rows in double, rounded once


def seq(xm, wm, acc=None):
    acc = np.zeros((xm.shape[0], wm.shape[1]), np.float32) if acc is None else acc
    for k in range(0, xm.shape[1], 2):
        # one MFMA step: two products join the accumulator (modelled as one float32 rounding of the exact three-term sum)
        acc = (acc.astype(np.float64) + xm[:, k:k + 1].astype(np.float64) * wm[k:k + 1].astype(np.float64)
               + xm[:, k + 1:k + 2].astype(np.float64) * wm[k + 1:k + 2].astype(np.float64)).astype(np.float32)
    return acc


def consumer(stored):
    return (scale32 * stored + beta32).astype(np.float32)


acc = seq(pix, W[256:])
report("seq (HIP path)", consumer((acc + addend).astype(np.float32)))
for C in (512, 256, 128):
    tot = None
    for k0 in range(0, 1024, C):
        part = seq(pix[:, k0:k0 + C], W[256 + k0:256 + k0 + C])
        tot = part if tot is None else (tot + part).astype(np.float32)
    report("chunk %d" % C, consumer((tot + addend).astype(np.float32)))
# what the yardstick would be with the products exact and ONE rounding of y, then BN in float32
y1 = (pool.astype(np.float64) @ W[:256].astype(np.float64) + pix.astype(np.float64) @ W[256:].astype(np.float64)).astype(np.float32)
report("y rounded once, BN in fp32", ((y1 - mm.astype(np.float32)) * scale32 + beta32).astype(np.float32))
report("y - mean rounded once", consumer(((pool.astype(np.float64) @ W[:256].astype(np.float64) + pix.astype(np.float64) @ W[256:].astype(np.float64)) - mm).astype(np.float32)))
