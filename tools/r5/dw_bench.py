"""Round 5: the stride-1 depthwise layers of the benchmarked plan (MobileNetV2 512x512, B=128) one by one, forward and
backward, on cold operands (two buffer sets), with an md5 of every output of the first launch (seeded inputs): two
libraries (DL3_LIBPATH) must print the same digests.
  python tools/r5/dw_bench.py [batch]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dl3_amd  # noqa: E402,F401
from dl3_amd import capi  # noqa: E402
from dl3_amd.capi import ptr  # noqa: E402

L = capi.lib()
ST = lambda: torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
md5 = lambda t: hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:10]


def timed(fn, reps=6):
    for i in range(2):
        fn(i % 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % 2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# H, W, C, rate, residual addend in the backward
CASES = [(64, 64, 960, 4, False), (64, 64, 576, 2, False), (64, 64, 384, 2, True), (64, 64, 192, 1, False), (128, 128, 144, 1, True),
         (256, 256, 32, 1, False)]
if B <= 16:   # Xception OS=8 (cfg4): the one-pixel forward kernel (rates that do not divide 32) and a 736-wide middle-flow layer
    CASES += [(64, 64, 2048, 12, False), (64, 64, 2048, 36, False), (64, 64, 736, 2, True)]
for H, W, C, r, has_add in CASES:
    rn = lambda *s: torch.randn(*s, device="cuda")
    xs, ys, gs, dxs = [rn(B, H, W, C) for _ in range(2)], [torch.empty(B, H, W, C, device="cuda") for _ in range(2)], \
        [rn(B, H, W, C) for _ in range(2)], [torch.empty(B, H, W, C, device="cuda") for _ in range(2)]
    adds = [rn(B, H, W, C) for _ in range(2)] if has_add else None
    w = rn(9, C) * 0.3
    v = [rn(C) for _ in range(7)]
    v[0], v[6] = torch.rand(C, device="cuda") + 0.5, torch.rand(C, device="cuda") + 0.5
    P = L.dl3_dwconv3x3_partials(B, H, W, C, 1, r, H, W, 0)
    part, dpart, wpart = torch.empty(P, C, 2, device="cuda"), torch.empty(P, C, 2, device="cuda"), torch.empty(P, 9, C, device="cuda")
    fwd = lambda i: capi.call("dl3_dwconv3x3_fwd", ptr(xs[i]), ptr(v[0]), ptr(v[1]), 2, ptr(w), ptr(ys[i]), B, H, W, C, 1, r, r, r, H, W,
                              ptr(part), 0, ST())
    bwd = lambda i: capi.call("dl3_dwconv3x3_bwd", ptr(gs[i]), ptr(ys[i]), ptr(v[2]), ptr(v[3]), ptr(v[4]), ptr(xs[i]), ptr(v[0]), ptr(v[1]),
                              2, ptr(w), ptr(dxs[i]), ptr(adds[i]) if has_add else None, ptr(v[5]), ptr(v[6]), ptr(dpart), ptr(wpart),
                              B, H, W, C, 1, r, r, r, H, W, 0, ST())
    fwd(0)
    d_f = md5(ys[0]) + " " + md5(part)
    bwd(0)
    d_b = md5(dxs[0]) + " " + md5(dpart) + " " + md5(wpart)
    e = B * H * W * C * 4.0
    tf, tb = timed(fwd), timed(bwd)
    print("dw %dx%dx%dx%d r%d add=%d  fwd %.3f ms %4.0f GB/s [%s] | bwd %.3f ms %4.0f GB/s [%s]" % (
        B, H, W, C, r, has_add, tf, 2 * e / tf / 1e6, d_f, tb, (4 + (1 if has_add else 0)) * e / tb / 1e6, d_b))
    del xs, ys, gs, dxs, adds
    torch.cuda.empty_cache()

# the stride-2 depthwise layers (dw_gather_fwd / dw_s2_bwd)
S2CASES = [(256, 256, 96), (128, 128, 144), (64, 64, 192)]
for H, W, C in S2CASES:
    Ho, Wo = H // 2, W // 2
    rn = lambda *s: torch.randn(*s, device="cuda")
    xs, ys = [rn(B, H, W, C) for _ in range(2)], [torch.empty(B, Ho, Wo, C, device="cuda") for _ in range(2)]
    w = rn(9, C) * 0.3
    v0, v1 = torch.rand(C, device="cuda") + 0.5, rn(C)
    P = L.dl3_dwconv3x3_partials(B, H, W, C, 2, 1, Ho, Wo, 0)
    part = torch.empty(P, C, 2, device="cuda")
    fwd = lambda i: capi.call("dl3_dwconv3x3_fwd", ptr(xs[i]), ptr(v0), ptr(v1), 2, ptr(w), ptr(ys[i]), B, H, W, C, 2, 1, 0, 0, Ho, Wo,
                              ptr(part), 0, ST())
    fwd(0)
    d_f = md5(ys[0]) + " " + md5(part)
    tf = timed(fwd)
    e = 4.0 * B * C * (H * W + Ho * Wo)
    print("dw-s2 %dx%dx%dx%d fwd %.3f ms %4.0f GB/s [%s]" % (B, H, W, C, tf, e / tf / 1e6, d_f))
    del xs, ys
    torch.cuda.empty_cache()
