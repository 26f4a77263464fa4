"""Round 6: single 1x1-convolution launches through the C ABI on cold operands (buffers rotate), one line per (shape, direction):
ms and TFLOP/s of the algorithmic FLOPs.  Kernel routes that are process-wide choices (environment read once) are compared by
running the script once per setting.

  python tools/r6/gemm_bench.py fwd:65536x736x736 bwd1:65536x736x736 bwd2:... wgrad:... wgraddy:...
    fwd   forward (producer BatchNorm + ReLU on load, BatchNorm sums out)
    bwd1  bwd-data, single-tensor dY, mask + sums against the forward input
    bwd2  bwd-data, two-tensor operand (g, y), mask + sums
    wgrad / wgraddy  weight gradient (two-tensor operand) without / with the dY store"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dl3_amd  # noqa: E402,F401
from dl3_amd import capi  # noqa: E402
from dl3_amd.capi import ptr  # noqa: E402

L = capi.lib()
ST = lambda: torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get("REPS", "8"))


def rnd(*shape):
    return torch.randn(*shape, device="cuda", dtype=torch.float32)


def timed(fn, nset):
    for i in range(min(2, nset)):
        fn(i % nset)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(REPS):
        fn(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


def run(kind, M, K, N):
    ns = int(max(2, min(4, np.ceil(1.5e9 / (4.0 * M * (K + 2 * N))))))
    xs = [rnd(M, K) for _ in range(ns)]
    w = rnd(K, N) * 0.1
    s, t = torch.rand(K, device="cuda") + 0.5, rnd(K) * 0.5
    if kind == "fwd":
        ys = [torch.empty(M, N, device="cuda") for _ in range(ns)]
        P = L.dl3_pwconv_partials(M, K, N)
        part = torch.empty(P, N, 2, device="cuda")
        fn = lambda i: capi.call("dl3_pwconv_fwd", ptr(xs[i]), K, ptr(s), ptr(t), 1, ptr(w), None, ptr(ys[i]), N, M, K, N, ptr(part), ST())
    elif kind in ("lfwd", "lbwd", "lwgrad"):
        # the logits layer (deeplabv3p.py:438): bias, no BatchNorm behind it, no mask in front of it
        bias = rnd(N)
        gs = [rnd(M, N) for _ in range(ns)]
        if kind == "lfwd":
            ys = [torch.empty(M, N, device="cuda") for _ in range(ns)]
            fn = lambda i: capi.call("dl3_pwconv_fwd", ptr(xs[i]), K, None, None, 0, ptr(w), ptr(bias), ptr(ys[i]), N, M, K, N, None, ST())
        elif kind == "lbwd":
            wT = w.t().contiguous()
            dxs = [torch.empty(M, K, device="cuda") for _ in range(ns)]
            fn = lambda i: capi.call("dl3_pwconv_bwd_data", ptr(gs[i]), N, None, N, None, None, None, ptr(wT), ptr(dxs[i]), K, None, K,
                                     None, None, 0, None, K, 1, 1.0, None, None, None, M, K, N, ST())
        else:
            nbytes = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
            ws = torch.empty(nbytes // 4 + 4, device="cuda")
            dw, db = torch.empty(K, N, device="cuda"), torch.empty(N, device="cuda")
            fn = lambda i: capi.call("dl3_pwconv_bwd_weight", ptr(xs[i]), K, None, None, 0, ptr(gs[i]), N, None, N, None, None, None,
                                     ptr(dw), ptr(db), M, K, N, ptr(ws), nbytes, ST())
    else:
        gs = [rnd(M, N) for _ in range(ns)]
        two = kind in ("bwd2", "wgrad", "wgraddy")
        yr = [rnd(M, N) for _ in range(ns)] if two else None
        cA, cB, cC = rnd(N), rnd(N), rnd(N)
        if kind.startswith("bwd"):
            wT = w.t().contiguous()
            dxs = [torch.empty(M, K, device="cuda") for _ in range(ns)]
            mean, invstd = rnd(K), torch.rand(K, device="cuda") + 0.5
            P = L.dl3_pwconv_partials(M, N, K)
            part = torch.empty(P, K, 2, device="cuda")
            fn = lambda i: capi.call("dl3_pwconv_bwd_data", ptr(gs[i]), N, ptr(yr[i]) if two else None, N, ptr(cA) if two else None,
                                     ptr(cB) if two else None, ptr(cC) if two else None, ptr(wT), ptr(dxs[i]), K, ptr(xs[i]), K,
                                     ptr(s), ptr(t), 1, None, K, 1, 1.0, ptr(mean), ptr(invstd), ptr(part), M, K, N, ST())
        else:
            nbytes = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
            ws = torch.empty(nbytes // 4 + 4, device="cuda")
            dy = torch.empty(M, N, device="cuda") if kind == "wgraddy" else None
            name = "dl3_pwconv_bwd_weight_dy" if dy is not None else "dl3_pwconv_bwd_weight"
            tail = (ptr(dy), N) if dy is not None else ()
            fn = lambda i: capi.call(name, ptr(xs[i]), K, ptr(s), ptr(t), 1, ptr(gs[i]), N, ptr(yr[i]), N, ptr(cA), ptr(cB), ptr(cC),
                                     None, None, M, K, N, ptr(ws), nbytes, *tail, ST())
    ms = timed(fn, ns)
    print("%-8s M=%7d K=%4d N=%4d  %7.3f ms  %6.1f TFLOP/s" % (kind, M, K, N, ms, 2.0 * M * K * N / ms / 1e9))
    sys.stdout.flush()


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        kind, dims = spec.split(":")
        M, K, N = [int(v) for v in dims.split("x")]
        run(kind, M, K, N)
        torch.cuda.empty_cache()
