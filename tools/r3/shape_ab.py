"""interleaved, repeated timing of a few GEMM shapes (fwd / dgrad / wgrad) in ONE process: the tuner's one-shot numbers
differ by up to 9 % between processes"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dl3_amd  # noqa: E402,F401
from dl3_amd import capi  # noqa: E402
from dl3_amd.capi import ptr  # noqa: E402

L = capi.lib()
st = torch.cuda.current_stream().cuda_stream
f = lambda *s: torch.randn(*s, device="cuda")
M = int(os.environ.get("AB_M", "65536"))
shapes = [tuple(int(q) for q in t.split("x")) for t in (sys.argv[1] if len(sys.argv) > 1 else "736x736,736x768,768x736,768x768,1024x1024").split(",")]
runs = {}
for K, N in shapes:
    a, b, c, sc, sh = f(M, K), f(K, N), f(M, N), f(K), f(K)
    pp = f(L.dl3_pwconv_partials(M, K, N), N, 2)
    runs[("fwd", K, N)] = (lambda a=a, b=b, c=c, sc=sc, sh=sh, pp=pp, K=K, N=N: capi.call(
        "dl3_pwconv_fwd", ptr(a), K, ptr(sc), ptr(sh), 2, ptr(b), None, ptr(c), N, M, K, N, ptr(pp), st), 2.0 * M * K * N)
    g, y, wT, dx, x = f(M, N), f(M, N), f(N, K), f(M, K), f(M, K)
    v = [f(max(K, N)) for _ in range(7)]
    pq = f(L.dl3_pwconv_partials(M, N, K), K, 2)
    runs[("dgrad", K, N)] = (lambda g=g, y=y, wT=wT, dx=dx, x=x, v=v, pq=pq, K=K, N=N: capi.call(
        "dl3_pwconv_bwd_data", ptr(g), N, ptr(y), N, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(wT), ptr(dx), K, ptr(x), K,
        ptr(v[3]), ptr(v[4]), 1, None, K, 1, 1.0, ptr(v[5]), ptr(v[6]), ptr(pq), M, K, N, st), 2.0 * M * K * N)
    dw = f(K, N)
    nb = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
    ws = torch.empty(nb // 4 + 4, device="cuda")
    runs[("wgrad", K, N)] = (lambda x=x, g=g, y=y, v=v, dw=dw, ws=ws, nb=nb, K=K, N=N: capi.call(
        "dl3_pwconv_bwd_weight", ptr(x), K, ptr(v[0]), ptr(v[1]), 1, ptr(g), N, ptr(y), N, ptr(v[2]), ptr(v[3]), ptr(v[4]),
        ptr(dw), None, M, K, N, ptr(ws), nb, st), 2.0 * M * K * N)
res = {k: [] for k in runs}
for k, (fn, _) in runs.items():
    for _ in range(3):
        fn()
for rep in range(5):
    for k, (fn, _) in runs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 10)
for k, ts in res.items():
    ts = sorted(ts)
    fl = runs[k][1]
    print("%-6s K=%4d N=%4d  ms min %.3f med %.3f max %.3f   TFLOP/s (med) %.1f" % (k[0], k[1], k[2], ts[0], ts[2], ts[-1], fl / ts[2] / 1e9))
