"""Times every tile configuration of the pointwise GEMM / weight-gradient kernels on the layer shapes of the
512x512 MobileNetV2 path (development aid; the forced config comes from DL3_GEMM_CFG / DL3_WGRAD_CFG).
usage: python tools/gemm_tune.py [--batch 16]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (pixels per image, K, N) of every distinct 1x1 conv of MobileNetV2/ASPP at 512x512
    (65536, 32, 16), (65536, 16, 96), (16384, 96, 24), (16384, 24, 144), (16384, 144, 24), (4096, 144, 32),
    (4096, 32, 192), (4096, 192, 32), (4096, 192, 64), (4096, 64, 384), (4096, 384, 64), (4096, 384, 96),
    (4096, 96, 576), (4096, 576, 96), (4096, 576, 160), (4096, 160, 960), (4096, 960, 160), (4096, 960, 320),
    (4096, 320, 256), (4096, 512, 256), (4096, 256, 21),
]


XCEPTION_SHAPES = [  # Xception OS=8 at 512x512
    (4096, 736, 736), (4096, 736, 1024), (4096, 1024, 1024), (4096, 1024, 1536), (4096, 1536, 1536), (4096, 1536, 2048),
    (4096, 2048, 256), (4096, 1280, 256), (4096, 256, 728), (16384, 256, 256), (16384, 304, 256), (16384, 128, 256),
    (65536, 128, 128), (65536, 64, 128), (65536, 288, 64),
]


def worker(batch, kind):
    import torch
    import dl3_amd  # noqa: F401
    from dl3_amd import capi
    from dl3_amd.capi import ptr
    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    f = lambda *s: torch.randn(*s, device="cuda")
    out = []
    for px, K, N in SHAPES:
        M = px * batch
        if kind == "fwd":
            a, b, c, sc, sh = f(M, K), f(K, N), f(M, N), f(K), f(K)
            pp = f(L.dl3_pwconv_partials(M, K, N), N, 2)
            run = lambda: capi.call("dl3_pwconv_fwd", ptr(a), K, ptr(sc), ptr(sh), 2, ptr(b), None, ptr(c), N, M, K, N, ptr(pp), st)
        elif kind == "dgrad":  # dx[M,K] = dY[M,N].WT with BN-backward operand, mask and stats
            g, y, wT, dx, x = f(M, N), f(M, N), f(N, K), f(M, K), f(M, K)
            v = [f(max(K, N)) for _ in range(7)]
            pp = f(L.dl3_pwconv_partials(M, N, K), K, 2)
            run = lambda: capi.call("dl3_pwconv_bwd_data", ptr(g), N, ptr(y), N, ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(wT),
                                    ptr(dx), K, ptr(x), K, ptr(v[3]), ptr(v[4]), 2, None, K, 1, 1.0, ptr(v[5]), ptr(v[6]),
                                    ptr(pp), M, K, N, st)
        else:
            x, g, y, dw = f(M, K), f(M, N), f(M, N), f(K, N)
            v = [f(max(K, N)) for _ in range(5)]
            nb = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
            ws = torch.empty(nb // 4 + 4, device="cuda")
            run = lambda: capi.call("dl3_pwconv_bwd_weight", ptr(x), K, ptr(v[0]), ptr(v[1]), 2, ptr(g), N, ptr(y), N,
                                    ptr(v[2]), ptr(v[3]), ptr(v[4]), ptr(dw), None, M, K, N, ptr(ws), nb, st)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10)
    print(" ".join("%.4f" % t for t in out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--worker", default=None)
    ap.add_argument("--xception", action="store_true")
    ap.add_argument("--shapes", default=os.environ.get("DL3_TUNE_SHAPES"), help="px x K x N, comma separated (overrides the table)")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--cfgs", default=None, help="comma list of forced configs to time next to auto (default: all)")
    args = ap.parse_args()
    if args.xception or os.environ.get("DL3_TUNE_XCEPTION"):
        SHAPES[:] = XCEPTION_SHAPES
        os.environ["DL3_TUNE_XCEPTION"] = "1"
    if args.shapes:
        SHAPES[:] = [tuple(int(q) for q in t.split("x")) for t in args.shapes.split(",")]
        os.environ["DL3_TUNE_SHAPES"] = args.shapes
    if args.worker:
        worker(args.batch, args.worker)
        sys.exit(0)
    for kind, var, ncfg in (("fwd", "DL3_GEMM_CFG", 7), ("dgrad", "DL3_GEMM_CFG", 7), ("wgrad", "DL3_WGRAD_CFG", 10)):
        if kind not in args.kinds.split(","):
            continue
        res = {}
        cfgs = list(range(ncfg)) if args.cfgs is None else [int(c) for c in args.cfgs.split(",")]
        for cfg in [-1] + cfgs:
            env = dict(os.environ)
            env[var] = str(cfg)
            r = subprocess.run([sys.executable, __file__, "--batch", str(args.batch), "--worker", kind], env=env,
                               capture_output=True, text=True)
            line = [l for l in r.stdout.strip().splitlines() if l and l[0].isdigit()]
            res[cfg] = [float(t) for t in line[-1].split()] if line else None
            if res[cfg] is None:
                print("cfg", cfg, "failed:", r.stderr[-300:])
        print("== %s (ms; auto | cfg0..): shape M/img,K,N" % kind)
        tot_auto = tot_best = 0.0
        for i, sh in enumerate(SHAPES):
            row = [res[c][i] if res[c] else float("nan") for c in [-1] + cfgs]
            best = min(range(1, len(row)), key=lambda j: row[j])
            tot_auto += row[0]
            tot_best += min(row[best], row[0])
            print("%-18s auto %.3f | %s | best cfg%d %.3f" % (sh, row[0], " ".join("%.3f" % t for t in row[1:]), cfgs[best - 1], row[best]))
        print("sum auto %.3f ms, sum best %.3f ms" % (tot_auto, tot_best))
