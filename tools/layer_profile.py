"""Per-launch timing of one training step (eager, HIP events around every libdl3 call) — a development aid.
usage: python tools/layer_profile.py [--batch 16] [--size 512] [--top 40]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl3_amd  # noqa: E402,F401
from dl3_amd import capi, graph as G  # noqa: E402
from dl3_amd.deeplabv3p import Deeplabv3  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--backbone", default="mobilenetv2")
ap.add_argument("--os", type=int, default=16)
ap.add_argument("--gemm", action="store_true", help="per-launch table of the 1x1-conv GEMMs against their roofline")
args = ap.parse_args()
G.clear_session(seed=1)
model = Deeplabv3(weights=None, input_shape=(args.size, args.size, 3), classes=21, backbone=args.backbone, OS=args.os)
eng = model._engine(args.batch, True, use_graph=False)
rng = np.random.default_rng(0)
eng.set_input(rng.integers(0, 256, (args.batch, args.size, args.size, 3)).astype(np.float32))
eng.set_targets(rng.integers(0, 22, (args.batch, args.size * args.size)).astype(np.float32))
for _ in range(2):
    eng.fwd_bwd()
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
rows = []
for phase, ops in (("fwd", eng.ops_fwd), ("bwd", eng.ops_bwd)):
    for name, fn, a, _ in ops:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            rc = fn(*a, st)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        dims = [x for x in a if isinstance(x, int) and 0 < x < 10 ** 7]
        rows.append((e0.elapsed_time(e1) / 3, phase, name, dims[-8:], a))
tot = sum(r[0] for r in rows)
print("total %.3f ms for %d launches (batch %d)" % (tot, len(rows), args.batch))
by = {}
for ms, ph, name, d, _ in rows:
    by[(ph, name)] = by.get((ph, name), 0) + ms
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print("  %-4s %-28s %8.3f ms %5.1f%%" % (k[0], k[1], v, 100 * v / tot))
print("top launches:")
for ms, ph, name, d, _ in sorted(rows, key=lambda r: -r[0])[:args.top]:
    print("  %7.3f ms %-4s %-26s %s" % (ms, ph, name, d))
if args.gemm:
    # M, K, N positions in the C signatures (include/dl3.h); bytes = operands read + result written
    pos = {"dl3_pwconv_fwd": (9, 10, 11), "dl3_pwconv_bwd_data": (22, 23, 24), "dl3_pwconv_bwd_weight": (14, 15, 16)}
    agg = {}
    for ms, ph, name, d, a in rows:
        if name not in pos:
            continue
        M, K, N = (a[i] for i in pos[name])
        two = a[2] is not None if name == "dl3_pwconv_bwd_data" else (a[7] is not None if name == "dl3_pwconv_bwd_weight" else False)
        if name == "dl3_pwconv_fwd":
            by_ = 4.0 * (M * K + M * N + K * N)
        elif name == "dl3_pwconv_bwd_data":
            by_ = 4.0 * (M * N * (2 if two else 1) + M * K * 2 + K * N)
        else:
            by_ = 4.0 * (M * K + M * N * (2 if two else 1) + K * N)
        key = (name[11:], M, K, N)
        t = agg.setdefault(key, [0, 0.0, by_])
        t[0] += 1
        t[1] += ms
    print("GEMM launches (grouped by shape): ms total | per launch | TFLOP/s | GB/s | roofline ms (157 TF, 6 TB/s) | frac")
    tg = tr = 0.0
    for (nm, M, K, N), (cnt, ms, by_) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * M * K * N
        roof = max(fl / 157.3e12, by_ / 6.0e12) * 1e3
        per = ms / cnt
        tg += ms
        tr += roof * cnt
        print("  %-10s M%-8d K%-5d N%-5d x%-2d %7.3f | %6.3f | %6.1f | %6.0f | %6.3f | %4.2f" %
              (nm, M, K, N, cnt, ms, per, fl / per / 1e9, by_ / per / 1e6, roof, roof / per))
    print("  GEMM total %.3f ms, roofline %.3f ms (%.2f)" % (tg, tr, tr / tg))
