"""PCIe-inclusive rate of the host-facing calls (Model.train_on_batch with host arrays), beside bench.py's resident-input
rate.  usage: python tools/pcie_bench.py [--batch 64]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl3_amd  # noqa
from dl3_amd import graph as G, utils as U
from dl3_amd.deeplabv3p import Deeplabv3
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=8)
a = ap.parse_args()
B = a.batch
G.clear_session(seed=1)
m = Deeplabv3(weights=None, input_shape=(512, 512, 3), classes=21)
m.compile(optimizer=dict(lr=7e-4, epsilon=1e-8, decay=1e-6))
rng = np.random.default_rng(0)
img8 = rng.integers(0, 256, (B, 512, 512, 3)).astype(np.uint8)
lab8 = rng.integers(0, 22, (B, 512, 512)).astype(np.uint8)
lab8[lab8 == 21] = 255
xf = img8.astype(np.float32)
yf = np.minimum(lab8, 21).astype(np.float32).reshape(B, -1, 1)
swf = (yf[:, :, 0] != 21).astype(np.float32)


def run(tag, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("%-64s %7.1f ms/step  %7.1f img/s" % (tag, dt * 1e3, B / dt))


run("float32 images + float32 Y/SW from host (the reference's arrays)", lambda: m.train_on_batch(xf, yf, swf))
run("uint8 images + uint8 label maps, targets prepared on the device", lambda: m.train_on_batch(img8, *U.prepare_targets(lab8, 21)))
eng = m._active
run("inputs resident (bench.py's timed region)", lambda: (eng.fwd_bwd(), eng.adam(None, 1.0)))
