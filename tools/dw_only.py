"""Launches only the roofline kernels (dilated depthwise fwd/bwd, B x 64 x 64 x 960, rate 4) — the target of the
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs, MI355X_MICROARCH.md §HBM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.cuda.set_device(0)
r = bench.roofline_leg(B)
print({k: round(v["ms"], 4) for k, v in r.items()})
