"""diagnostic (round 5): every gradient tensor of the 72x56 MobileNetV2 training step of
tests/test_gpu_model.py::test_odd_and_non_square_inputs against the float64 oracle, in graph order."""
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

shape, classes, B = (72, 56, 3), 3, 2
G.clear_session()
model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone="mobilenetv2", OS=16)
params = O.init_params(O.param_shapes("mobilenetv2", classes, head="deeplab"), seed=1)
rng = np.random.default_rng(shape[0])
x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
params = O.calibrate_bn(params, x, **kw)
for l in model.layers:
    if l.weights:
        l.set_weights([params[n] for n in l.weights])
p64 = {k: v.astype(np.float64) for k, v in params.items()}
ref, _ = O.forward(p64, x.astype(np.float64), **kw)
model.predict(x, batch_size=B)
labels = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
sw = (labels < classes).astype(np.float32)
eng = model._engine(B, True, dropout=False, use_graph=False)
eng.set_input(x)
eng.set_targets(labels, sw)
eng.fwd_bwd()
torch.cuda.synchronize()
loss, grads, logits, _ = O.train_grads(p64, x.astype(np.float64), labels.astype(np.float64), sw.astype(np.float64), **kw)
l2 = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) / (np.linalg.norm(b) + 1e-300))
print("logits rel %.3e  loss gpu %.8f oracle %.8f" % (l2(eng.logits(), logits), float(eng.loss[0].item()), loss))
for l in model.layers:
    for n in l.weights:
        if "/moving_" in n or grads.get(n) is None:
            continue
        print("%-52s %.3e" % (n, l2(eng.grad_of(n), grads[n])))
