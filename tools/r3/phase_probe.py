"""Probe (round 3): where does a row tile of the stream GEMM spend its time?  Runs single launches of the layer shapes
through a -DDL3_PHASE_TIMING build of libdl3.so (build_variants/libdl3_timing.so) whose workgroups add up clock64()
deltas of their prologue (tile start -> first K-tile ready), K loop and epilogue per wave."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DL3_LIBPATH"] = os.path.join(ROOT, "build_variants", os.environ.get("PROBE_LIB", "libdl3_timing.so"))
import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dl3_amd  # noqa: E402,F401
from dl3_amd import capi  # noqa: E402
from dl3_amd.capi import ptr  # noqa: E402

L = capi.lib()
L.dl3_debug_phase_buffer.argtypes = [ctypes.c_void_p]
if os.environ.get("PROBE_MATH") == "split":
    capi.set_gemm_math("split")
st = torch.cuda.current_stream().cuda_stream
f = lambda *s: torch.randn(*s, device="cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(4096, 64, 384), (4096, 384, 64), (4096, 96, 576), (4096, 576, 96), (4096, 160, 960), (4096, 960, 160),
          (4096, 960, 320), (16384, 24, 144), (16384, 144, 24), (65536, 16, 96)]
if os.environ.get("PROBE_SHAPES"):
    SHAPES = [tuple(int(q) for q in t.split("x")) for t in os.environ["PROBE_SHAPES"].split(",")]
KINDS = os.environ.get("PROBE_KINDS", "fwd,dgrad").split(",")
dbg = torch.zeros(8192 * 32, dtype=torch.int64, device="cuda")
print("%-22s %-6s %8s | per row tile and wave, microseconds at the measured clock: prologue  K-loop  epilogue | tiles/wg  wgs  t_mfma/tile" % ("shape (px,K,N)", "kind", "ms"))
for px, K, N in SHAPES:
    M = px * B
    for kind in KINDS:
        if kind == "fwd":
            a, b, c, sc, sh = f(M, K), f(K, N), f(M, N), f(K), f(K)
            pp = f(L.dl3_pwconv_partials(M, K, N), N, 2)
            run = lambda: capi.call("dl3_pwconv_fwd", ptr(a), K, ptr(sc), ptr(sh), 2, ptr(b), None, ptr(c), N, M, K, N, ptr(pp), st)
            red, outw = K, N
        elif kind == "wgrad":
            x, g, y, dw = f(M, K), f(M, N), f(M, N), f(K, N)
            v = [f(max(K, N)) for _ in range(5)]
            nb = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
            ws = torch.empty(nb // 4 + 4, device="cuda")
            run = lambda: capi.call("dl3_pwconv_bwd_weight", ptr(x), K, ptr(v[0]), ptr(v[1]), 2, ptr(g), N, ptr(y), N,
                                    ptr(v[2]), ptr(v[3]), ptr(v[4]), ptr(dw), None, M, K, N, ptr(ws), nb, st)
            red, outw = M, N
        else:
            g, y, wT, dx, x = f(M, N), f(M, N), f(N, K), f(M, K), f(M, K)
            v = [f(max(K, N)) for _ in range(7)]
            pp = f(L.dl3_pwconv_partials(M, N, K), K, 2)
            run = lambda: capi.call("dl3_pwconv_bwd_data", ptr(g), N, ptr(y), N, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(wT),
                                    ptr(dx), K, ptr(x), K, ptr(v[3]), ptr(v[4]), 2, None, K, 1, 1.0, ptr(v[5]), ptr(v[6]),
                                    ptr(pp), M, K, N, st)
            red, outw = N, K
        L.dl3_debug_phase_buffer(None)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        dbg.zero_()
        L.dl3_debug_phase_buffer(dbg.data_ptr())
        run()
        torch.cuda.synchronize()
        L.dl3_debug_phase_buffer(None)
        d = dbg.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 3] > 0]
        nw = d.shape[0]
        tot = d[:, :3].sum(1).astype(np.float64)
        clk = tot.mean() / (ms * 1e-3)  # cycles per second, assuming a wave is busy for the whole launch
        per = d[:, :3].sum(0) / d[:, 3].sum() / clk * 1e6
        # fp32 MFMA time of one tile on one SIMD: 64 cycles per 32x32x2 MFMA at 2.4 GHz
        if kind == "wgrad":
            # d[1] = whole kernel cycles of the wave, d[3] = stages of 16 pixel rows, d[4] / d[5] = waits
            tot = d[:, 1].astype(np.float64)
            clk = tot.mean() / (ms * 1e-3)
            st_us = d[:, 1].sum() / d[:, 3].sum() / clk * 1e6
            wv = d[:, 4:6].sum(0) / d[:, 3].sum() / clk * 1e6
            print("%-22s %-6s %8.3f | per 16-row stage and wave: %.3f us (MFMA time of one wave at 2.4 GHz: see tile), waiting for the loads %.3f us, at the barrier %.3f us | stages/wg %.0f  wgs %d  clk %.2f GHz" % (
                (px, K, N), kind, ms, st_us, wv[0], wv[1], d[:, 3].mean(), nw // 4, clk / 1e9))
            continue
        wv = d[:, 4:6].sum(0) / d[:, 3].sum() / clk * 1e6
        print("%-22s %-6s %8.3f | %8.2f %8.2f %8.2f | %6.1f %6d   clk %.2f GHz | in the K loop: waiting for the operand loads %.2f us, at the barrier %.2f us per tile" % (
            (px, K, N), kind, ms, per[0], per[1], per[2], d[:, 3].mean(), nw // 4, clk / 1e9, wv[0], wv[1]))
