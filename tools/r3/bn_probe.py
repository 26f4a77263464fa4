"""Diagnostic (round 3): is the training-mode forward of Xception further from float64 than torch-fp32 because of the
BatchNorm batch statistics?  Runs the forward plan launch by launch; in 'exact' mode the partial rows every
dl3_bn_finalize folds are replaced by the float64 column sums of the very tensor the convolution wrote (split hi/lo over
two partial rows, so nothing is lost in the float storage).  torch is used here as a measuring instrument only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dl3_amd  # noqa: E402,F401
from dl3_amd import graph as G  # noqa: E402
from dl3_amd.deeplabv3p import Deeplabv3  # noqa: E402
from oracle import dl3_oracle as O  # noqa: E402
from oracle import torch_ref as T  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
backbone = sys.argv[2] if len(sys.argv) > 2 else "xception"
OS = 8 if backbone == "xception" else 16
shape, classes, B = (size, size, 3), 21, 2
torch.set_num_threads(32)
G.clear_session()
model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone=backbone, OS=OS)
params = O.init_params(O.param_shapes(backbone, classes), seed=1)
for l in model.layers:
    if l.weights:
        l.set_weights([params[n] for n in l.weights])
rng = np.random.default_rng(0)
x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
labels = rng.integers(0, classes + 1, (B, size * size)).astype(np.float32)
sw = ((labels < classes) * rng.uniform(0.5, 2.0, labels.shape)).astype(np.float32)
kw = dict(backbone=backbone, input_shape=shape, classes=classes, OS=OS, head="deeplab")
loss, grads, logits = T.train_grads(params, x, labels, sw, dtype=torch.float64, **kw)
loss32, grads32, logits32 = T.train_grads(params, x, labels, sw, dtype=torch.float32, **kw)
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
print("torch fp32 vs fp64 logits rel err %.3e" % rel(logits32, logits))

eng = model._engine(B, True, dropout=False, use_graph=False)
eng.set_input(x)
eng.set_targets(labels, sw)
st = torch.cuda.current_stream().cuda_stream
sites = {i: (l, buf, off, C, unit) for i, l, buf, off, C, unit in eng.bn_sites}


def run(mode):
    eng._prep()
    worst = []
    for i, (name, fn, args, _) in enumerate(eng.ops_fwd):
        if i in sites:
            l, buf, off, C, unit = sites[i]
            y = buf.t.view(buf.M, buf.ld)[:, off:off + C].double()
            s1, s2 = y.sum(0), (y * y).sum(0)
            P = unit.P
            part = unit.stat.view(P, C, 2)
            g1, g2 = part[:, :, 0].double().sum(0), part[:, :, 1].double().sum(0)
            n = float(buf.M)
            var_e = s2 / n - (s1 / n) ** 2
            var_g = g2 / n - (g1 / n) ** 2
            rv = ((var_g - var_e).abs() / var_e.clamp_min(1e-30))
            ratio = ((s1 / n) ** 2 / var_e.clamp_min(1e-30))
            worst.append((float(rv.max()), float(rv.median()), float(ratio.max()), l.name, P))
            if mode == "exact":
                part.zero_()
                h1, h2 = s1.float(), s2.float()
                part[0, :, 0], part[0, :, 1] = h1, h2
                if P > 1:
                    part[1, :, 0], part[1, :, 1] = (s1 - h1.double()).float(), (s2 - h2.double()).float()
        rc = fn(*args, st)
        assert rc == 0, name
    torch.cuda.synchronize()
    got = eng.logits()
    print("mode %-6s: gpu logits rel err vs float64 %.3e" % (mode, rel(got, logits)))
    return worst


w = run("asis")
w.sort(reverse=True)
print("largest relative variance errors of the fp32 partial sums (max over channels, median, max mean^2/var, layer, P):")
for r in w[:12]:
    print("   %.2e  %.2e  %9.1f  %-44s P=%d" % r)
print("median over layers of the max-over-channels variance error: %.2e" % float(np.median([r[0] for r in w])))
run("exact")
