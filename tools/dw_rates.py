"""Times the dilated depthwise forward/backward kernels over the rates the path uses (development aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl3_amd  # noqa
from dl3_amd import capi
from dl3_amd.capi import ptr
L = capi.lib()
st = torch.cuda.current_stream().cuda_stream
f = lambda *s: torch.randn(*s, device="cuda")


def time_kernel(launch, iters=20, warmup=3):
    for _ in range(warmup):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cases = [(32, 64, 64, 960, 4), (32, 64, 64, 576, 2), (8, 64, 64, 2048, 12), (8, 64, 64, 2048, 24), (8, 64, 64, 2048, 36), (8, 64, 64, 728, 2), (8, 64, 64, 1536, 4),
         (16, 64, 64, 960, 4), (16, 64, 64, 576, 2), (16, 64, 64, 192, 1), (16, 128, 128, 144, 1), (16, 256, 256, 32, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "align":
    # does a pixel row that is not a whole number of 128-byte lines cost bandwidth?  (xception middle flow: 728 channels)
    cases = [(16, 64, 64, c, 2) for c in (704, 728, 736, 768, 1024)] + [(16, 128, 128, c, 1) for c in (128, 144, 160)]
elif len(sys.argv) > 1:
    cases = cases[:int(sys.argv[1])]
for N, H, W, C, r in cases:
    x, w, y, g, dx = f(N, H, W, C), f(9, C), f(N, H, W, C), f(N, H, W, C), f(N, H, W, C)
    v = [f(C) for _ in range(7)]
    P = L.dl3_dwconv3x3_partials(N, H, W, C, 1, r, H, W, 0)
    part, dpart, wpart = f(P, C, 2), f(P, C, 2), f(P, 9, C)
    fwd = lambda: capi.call("dl3_dwconv3x3_fwd", ptr(x), ptr(v[0]), ptr(v[1]), 2, ptr(w), ptr(y), N, H, W, C, 1, r, r, r, H, W, ptr(part), 0, st)
    bwd = lambda: capi.call("dl3_dwconv3x3_bwd", ptr(g), ptr(y), ptr(v[2]), ptr(v[3]), ptr(v[4]), ptr(x), ptr(v[0]), ptr(v[1]), 2, ptr(w),
                            ptr(dx), None, ptr(v[5]), ptr(v[6]), ptr(dpart), ptr(wpart), N, H, W, C, 1, r, r, r, H, W, 0, st)
    e = N * H * W * C * 4.0
    tf, tb = time_kernel(fwd), time_kernel(bwd)
    print("N%d %dx%dx%d r%-2d P=%5d  fwd %.3f ms %.2f TB/s | bwd %.3f ms %.2f TB/s" % (N, H, W, C, r, P, tf, 2 * e / tf / 1e9, tb, 4 * e / tb / 1e9))
