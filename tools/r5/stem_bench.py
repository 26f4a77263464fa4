"""Round 5: the MobileNetV2 stem convolution (deeplabv3p.py:318, 3 -> 32 channels, stride 2) of the benchmarked plan alone:
forward and weight gradient at 128 x 512 x 512, cold operands (two buffer sets).  DL3_LIBPATH selects the library.
  python tools/r5/stem_bench.py [batch]"""
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
H = W = 512
Ho = Wo = 256
xs = [torch.randint(0, 256, (B, H, W, 3), device="cuda").float() for _ in range(2)]
ys = [torch.empty(B, Ho, Wo, 32, device="cuda") for _ in range(2)]
gs = [torch.randn(B, Ho, Wo, 32, device="cuda") for _ in range(2)]
w = torch.randn(3, 3, 3, 32, device="cuda") * 0.1
sc, sh = torch.full((3,), 1 / 127.5, device="cuda"), torch.full((3,), -1.0, device="cuda")
cA, cB, cC = torch.randn(32, device="cuda"), torch.randn(32, device="cuda"), torch.randn(32, device="cuda")
P = L.dl3_conv3x3_partials(B, Ho, Wo, 32)
part = torch.empty(P, 32, 2, device="cuda")
wpart = torch.empty(P, 27, 32, device="cuda")


def timed(fn, reps=10):
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


fwd = lambda i: capi.call("dl3_conv3x3_fwd", ptr(xs[i]), ptr(sc), ptr(sh), 0, ptr(w), ptr(ys[i]), B, H, W, 3, 32, 2, 0, 0, Ho, Wo,
                          ptr(part), ST())
wg = lambda i: capi.call("dl3_conv3x3_bwd_weight", ptr(xs[i]), ptr(sc), ptr(sh), 0, ptr(gs[i]), ptr(ys[i]), ptr(cA), ptr(cB), ptr(cC),
                         ptr(wpart), B, H, W, 3, 32, 2, 0, 0, Ho, Wo, ST())
by_f = 4.0 * (B * H * W * 3 + B * Ho * Wo * 32)
by_w = 4.0 * (B * H * W * 3 + 2 * B * Ho * Wo * 32)
ms = timed(fwd)
import hashlib  # noqa: E402
print("stem fwd   B=%d  %.3f ms  %.0f GB/s (input once + output)   y md5 %s  partial sums md5 %s" % (
    B, ms, by_f / ms / 1e6, hashlib.md5(ys[0].cpu().numpy().tobytes()).hexdigest()[:12], hashlib.md5(part.cpu().numpy().tobytes()).hexdigest()[:12]))
ms = timed(wg)
print("stem wgrad B=%d  %.3f ms  %.0f GB/s (input once + g + y)" % (B, ms, by_w / ms / 1e6))
