"""Round 5: the HBM-bound 1x1-convolution launches of the benchmarked plan (MobileNetV2 512x512, B=128) one by one, on
cold operands (the buffers of a shape rotate so that no launch re-reads what the previous one left in L2 / MALL).

  python tools/r5/pw_hbm_bench.py [fwd] [fused]

forward: dl3_pwconv_fwd as the engine issues it (producer's BatchNorm + ReLU6 on load, BatchNorm partial sums out).  The
weight-stationary route is a process-wide choice (DL3_FWD_WS=0|1): run the script once per setting.
fused:   dl3_pwconv_bwd_fused, round-4 kernel (DL3_FUSED_V=1) against round 5, in one process.
Prints one line per (shape, variant): ms, algorithmic GB/s (the formulas of bench.py's in-situ table)."""
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
REPS = int(os.environ.get("REPS", "6"))


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


def nsets(bytes_per_launch):
    return int(max(2, min(6, np.ceil(1.2e9 / bytes_per_launch))))


FWD = [(8388608, 32, 16), (8388608, 16, 96), (2097152, 96, 24), (2097152, 24, 144), (2097152, 144, 24), (524288, 144, 32),
       (524288, 32, 192), (524288, 192, 32), (524288, 192, 64), (524288, 64, 384)]


def bench_fwd():
    print("# forward  DL3_FWD_WS=%s" % os.environ.get("DL3_FWD_WS", "(default: on)"))
    for M, K, N in FWD:
        by = 4.0 * (M * K + M * N + K * N)
        ns = nsets(by)
        xs = [rnd(M, K) for _ in range(ns)]
        ys = [torch.empty(M, N, device="cuda") for _ in range(ns)]
        w, s, t = rnd(K, N) * 0.2, torch.rand(K, device="cuda") + 0.5, rnd(K) * 0.5
        P = L.dl3_pwconv_partials(M, K, N)
        part = torch.empty(P, N, 2, device="cuda")

        def fn(i):
            capi.call("dl3_pwconv_fwd", ptr(xs[i]), K, ptr(s), ptr(t), 2, ptr(w), None, ptr(ys[i]), N, M, K, N, ptr(part), ST())

        ms = timed(fn, ns)
        print("fwd   M=%8d K=%3d N=%3d impl=%d  %7.3f ms  %6.0f GB/s" % (M, K, N, L.dl3_pwconv_fwd_impl(M, K, N), ms, by / ms / 1e6))
        del xs, ys
        torch.cuda.empty_cache()


# M, K, N, residual addend, sums against another tensor — as the engine lowers MobileNetV2's first blocks
FUSED = [(8388608, 32, 16, False, False), (8388608, 16, 96, False, False), (2097152, 96, 24, False, False),
         (2097152, 24, 144, False, False), (2097152, 24, 144, True, True), (2097152, 144, 24, False, False),
         (524288, 144, 32, False, False), (524288, 32, 192, False, False), (524288, 32, 192, True, True),
         (524288, 192, 32, False, False)]


def bench_fused():
    for M, K, N, has_add, foreign in FUSED:
        by = 4.0 * (M * K + 2 * M * N + K * N + M * K * (1 + (1 if has_add else 0) + (1 if foreign else 0)))
        ns = nsets(by)
        xs, gs, ysr, dxs = [rnd(M, K) for _ in range(ns)], [rnd(M, N) for _ in range(ns)], [rnd(M, N) for _ in range(ns)], \
            [torch.empty(M, K, device="cuda") for _ in range(ns)]
        adds = [rnd(M, K) for _ in range(ns)] if has_add else None
        sxs = [rnd(M, K) for _ in range(ns)] if foreign else None
        wT = rnd(N, K) * 0.2
        cA, cB, cC = rnd(N), rnd(N), rnd(N)
        s, t = torch.rand(K, device="cuda") + 0.5, rnd(K) * 0.5
        mean, invstd = rnd(K), torch.rand(K, device="cuda") + 0.5
        act = 0 if has_add else 2
        for label, env in (("r4", {"DL3_FUSED_V": "1"}), ("r5", {})):
            os.environ.pop("DL3_FUSED_V", None)
            os.environ.update(env)
            if not L.dl3_pwconv_bwd_fused_supported(M, K, N):
                print("fused M=%8d K=%3d N=%3d add=%d foreign=%d %-10s  unsupported" % (M, K, N, has_add, foreign, label))
                continue
            S = L.dl3_pwconv_bwd_fused_splits(M, K, N)
            nbytes = L.dl3_pwconv_bwd_fused_workspace(M, K, N)
            ws, part = torch.empty(S * K * N + 4, device="cuda"), torch.empty(S, K, 2, device="cuda")

            def fn(i):
                sx = sxs[i] if foreign else xs[i]
                capi.call("dl3_pwconv_bwd_fused", ptr(xs[i]), K, ptr(s) if act else None, ptr(t) if act else None, act, ptr(gs[i]), N,
                          ptr(ysr[i]), N, ptr(cA), ptr(cB), ptr(cC), ptr(wT), None, ptr(dxs[i]), K,
                          ptr(adds[i]) if has_add else None, K, ptr(sx), K, ptr(mean), ptr(invstd), ptr(part), M, K, N, ptr(ws),
                          nbytes, ST())

            ms = timed(fn, ns)
            print("fused M=%8d K=%3d N=%3d add=%d foreign=%d %-10s  %7.3f ms  %6.0f GB/s" % (
                M, K, N, has_add, foreign, label, ms, by / ms / 1e6))
        os.environ.pop("DL3_FUSED_V", None)
        del xs, gs, ysr, dxs, adds, sxs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    what = sys.argv[1:] or ["fwd", "fused"]
    if "fwd" in what:
        bench_fwd()
    if "fused" in what:
        bench_fused()
