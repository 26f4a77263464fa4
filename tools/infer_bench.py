"""Forward-only (inference engine, frozen BN, no dropout) throughput of Deeplabv3().predict's device part.
usage: python tools/infer_bench.py [--batch 32] [--backbone mobilenetv2] [--os 16]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl3_amd  # noqa
from dl3_amd import graph as G
from dl3_amd.deeplabv3p import Deeplabv3
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--backbone", default="mobilenetv2")
ap.add_argument("--os", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
G.clear_session(seed=1)
m = Deeplabv3(weights=None, input_shape=(512, 512, 3), classes=21, backbone=a.backbone, OS=a.os)
eng = m._engine(a.batch, False)
eng.set_input(np.random.default_rng(0).integers(0, 256, (a.batch, 512, 512, 3)).astype(np.float32))
for _ in range(3):
    eng.forward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    eng.forward()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print("%s OS=%d B=%d forward (logits at full resolution): %.2f ms/step, %.0f img/s, %d launches" %
      (a.backbone, a.os, a.batch, dt * 1e3, a.batch / dt, len(eng.ops_fwd)))
