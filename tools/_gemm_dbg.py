import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dl3_amd
from dl3_amd import capi
from dl3_amd.capi import ptr
L = capi.lib()
st = torch.cuda.current_stream().cuda_stream
f = lambda *s: torch.randn(*s, device="cuda")
dbg = torch.zeros(64 * 4 * 8 * 4, dtype=torch.int64, device="cuda")
for (M, K, N, kind) in [(131072, 160, 960, "fwd"), (131072, 960, 160, "fwd"), (131072, 160, 960, "dgrad"), (131072, 960, 160, "dgrad")]:
    if kind == "fwd":
        a, b, c, sc, sh = f(M, K), f(K, N), f(M, N), f(K), f(K)
        pp = f(L.dl3_pwconv_partials(M, K, N), N, 2)
        run = lambda: capi.call("dl3_pwconv_fwd", ptr(a), K, ptr(sc), ptr(sh), 2, ptr(b), None, ptr(c), N, M, K, N, ptr(pp), st)
    else:
        g, y, wT, dx, x = f(M, N), f(M, N), f(N, K), f(M, K), f(M, K)
        v = [f(max(K, N)) for _ in range(7)]
        pp = f(L.dl3_pwconv_partials(M, N, K), K, 2)
        run = lambda: capi.call("dl3_pwconv_bwd_data", ptr(g), N, ptr(y), N, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(wT),
                                ptr(dx), K, ptr(x), K, ptr(v[3]), ptr(v[4]), 2, None, K, 1, 1.0, ptr(v[5]), ptr(v[6]), ptr(pp), M, K, N, st)
    os.environ.pop("DL3_GEMM_DBG_PTR", None)
    for _ in range(3): run()
    torch.cuda.synchronize()
    os.environ["DL3_GEMM_DBG_PTR"] = hex(dbg.data_ptr())
    dbg.zero_()
    run()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(64, 4, 8, 4).astype(np.float64)
    ok = d[..., 3] > 0
    pro = (d[..., 1] - d[..., 0])[ok]; main = (d[..., 2] - d[..., 1])[ok]; epi = (d[..., 3] - d[..., 2])[ok]
    # gap between consecutive tiles of a wave (epilogue end -> next tile start)
    nxt = d[:, :, 1:, 0] - d[:, :, :-1, 3]
    gap = nxt[(d[:, :, 1:, 3] > 0) & (d[:, :, :-1, 3] > 0)]
    print("%s M%d K%d N%d: tiles timed %d | prologue %.0f  main %.0f  epilogue %.0f  inter-tile gap %.0f  (mean clock64 ticks; per K-tile %.0f)" %
          (kind, M, K, N, ok.sum(), pro.mean(), main.mean(), epi.mean(), gap.mean() if gap.size else -1, main.mean() / ((K if kind == "fwd" else N) / 16)))
