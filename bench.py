"""bench.py — images/sec forward+backward(+Adam) of the DeepLabV3+ MobileNetV2 path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N=1: plain python; N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one resident batch: forward, loss, backward (one replayed hipGraph of
libdl3.so launches), RCCL all-reduce of the flat gradient arena (N>1), Adam.  Inputs are synthetic and already in
HBM when the timed region starts.  Workload = BASELINE.json configs[1]: Deeplabv3(backbone='mobilenetv2',
input_shape=(512,512,3), classes=21, OS=16), fp32, BatchNorm in training (batch-statistics) mode, dropout on.
Rank 0 prints ONE JSON line with `roofline` (dominant HBM-bound kernel: the dilated depthwise 3x3) and, at N=1,
`cpu_baseline` (the CPU restatement of the same step timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import dl3_amd  # noqa: E402,F401
from dl3_amd import capi, graph as G  # noqa: E402
from dl3_amd.capi import ptr  # noqa: E402
from dl3_amd.deeplabv3p import Deeplabv3  # noqa: E402
from dl3_amd.parallel import DataParallel  # noqa: E402
from dl3_amd.utils import SegModel  # noqa: E402

T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # fp32 MFMA = fp32 vector peak


def build_engine(args):
    G.clear_session(seed=1)
    shape = (args.size, args.size, 3)
    if args.head == "deeplab":
        model = Deeplabv3(weights=None, input_shape=shape, classes=21, backbone=args.backbone, OS=args.os)
    else:
        model = SegModel(image_size=shape[:2]).create_seg_model(args.head, n=21, backbone=args.backbone)
    eng = model._engine(args.batch, True, bn_mode=args.bn_mode, dropout=True, use_graph=not args.no_graph)
    rng = np.random.default_rng(1000 + int(os.environ.get("RANK", "0")))
    x = rng.integers(0, 256, (args.batch,) + shape).astype(np.float32)
    y = rng.integers(0, 22, (args.batch, args.size * args.size)).astype(np.float32)  # 21 = void
    eng.set_input(x)
    eng.set_targets(y)
    return model, eng


def time_kernel(launch, iters=20, warmup=3):
    """average device time (ms) of one launch, HIP events on the stream the kernel is launched on"""
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


def roofline_leg(B):
    """The dominant HBM-bound kernel family of the path: dilated depthwise 3x3 (rate 4, 64x64x960 =
    expanded_conv_14/15/16_depthwise).  Algorithmic bytes per launch = 4*(in + out + w) = 4*(2*B*64*64*960 + 9*960)
    (SURVEY §8d 'K1 fwd').  Also reports the fused backward and the two large pointwise GEMMs (fp32 MFMA)."""
    L = capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    N, H, W, C, r = B, 64, 64, 960, 4
    f = lambda *s: torch.randn(*s, device="cuda", dtype=torch.float32)
    x, w, y, g, dx = f(N, H, W, C), f(9, C), f(N, H, W, C), f(N, H, W, C), f(N, H, W, C)
    vec = [f(C) for _ in range(7)]
    P = L.dl3_dwconv3x3_partials(N, H, W, C, 1, r, H, W, 0)
    part, dpart, wpart = f(P, C, 2), f(P, C, 2), f(P, 9, C)
    out = {}

    def fwd():
        capi.call("dl3_dwconv3x3_fwd", ptr(x), ptr(vec[0]), ptr(vec[1]), 2, ptr(w), ptr(y), N, H, W, C, 1, r, r, r, H, W,
                  ptr(part), 0, st)

    def bwd():
        capi.call("dl3_dwconv3x3_bwd", ptr(g), ptr(y), ptr(vec[2]), ptr(vec[3]), ptr(vec[4]), ptr(x), ptr(vec[0]),
                  ptr(vec[1]), 2, ptr(w), ptr(dx), None, ptr(vec[5]), ptr(vec[6]), ptr(dpart), ptr(wpart), N, H, W, C, 1,
                  r, r, r, H, W, 0, st)

    elems = N * H * W * C
    ms = time_kernel(fwd)
    bytes_f = 4.0 * (2 * elems + 9 * C)
    out["dw_r4_fwd"] = dict(ms=ms, bytes=bytes_f, gbs=bytes_f / ms / 1e6)
    ms = time_kernel(bwd)
    bytes_b = 4.0 * (4 * elems + 2 * 9 * C)  # reads g, yraw, x; writes dx (BN-backward on load needs yraw)
    out["dw_r4_bwd"] = dict(ms=ms, bytes=bytes_b, gbs=bytes_b / ms / 1e6)
    # pointwise GEMMs of the same block: expand 160->960 and project 960->160, M = B*4096
    M = N * H * W
    for name, K, Nn in (("pw_expand_160_960", 160, 960), ("pw_project_960_160", 960, 160)):
        a, b, c = f(M, K), f(K, Nn), f(M, Nn)
        sc, sh = f(K), f(K)
        Pp = L.dl3_pwconv_partials(M, K, Nn)
        pp = f(Pp, Nn, 2)

        def gemm():
            capi.call("dl3_pwconv_fwd", ptr(a), K, ptr(sc), ptr(sh), 2, ptr(b), None, ptr(c), Nn, M, K, Nn, ptr(pp), st)

        ms = time_kernel(gemm)
        out[name] = dict(ms=ms, tflops=2.0 * M * K * Nn / ms / 1e9, gbs=4.0 * (M * K + M * Nn + K * Nn) / ms / 1e6)
    return out


def cpu_baseline_leg(args):
    """CPU restatement of the same training step (oracle/torch_ref.py: torch/oneDNN on the host cores) on a bounded
    sample.  kind='port': the reference's own Keras/TensorFlow CPU path cannot be installed here."""
    from oracle import dl3_oracle as O
    from oracle import torch_ref as T
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # oneDNN on a 2-image batch stops scaling (and collapses from oversubscription) far below a 256-thread host:
    # use at most 32 threads and report that count
    cores = min(avail, args.cpu_threads)
    torch.set_num_threads(cores)
    log("cpu baseline on %d of %d host cores" % (cores, avail))
    B = 2
    kw = dict(backbone=args.backbone, input_shape=(args.size, args.size, 3), classes=21, OS=16)
    params = O.init_params(O.param_shapes(args.backbone, 21), seed=1)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (B, args.size, args.size, 3)).astype(np.float32)
    y = rng.integers(0, 22, (B, args.size, args.size)).astype(np.float32)
    w = (y < 21).astype(np.float32)
    T.train_grads(params, x, y, w, **kw)  # warm-up
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        T.train_grads(params, x, y, w, **kw)
    dt = time.perf_counter() - t0
    return dict(value=B * steps / dt, unit="img/s", cores=cores, kind="port",
                sample="%d steps x %d images %dx%d fwd+bwd, torch-CPU (oneDNN) restatement oracle/torch_ref.py, "
                       "after 1 warm-up step" % (steps, B, args.size, args.size))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64,
                    help="images per GPU (4x SegModel.batch_size, utils.py:162; 50 GB of the 288 GB HBM; throughput by "
                         "batch on one MI355X: 32 -> 1002, 64 -> 1057, 128 -> 1097 img/s)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--backbone", default="mobilenetv2")
    ap.add_argument("--os", type=int, default=16, help="output stride (Xception only; MobileNetV2 always runs at 8)")
    ap.add_argument("--head", default="deeplab", choices=["deeplab", "original", "subpixel"])
    ap.add_argument("--bn-mode", default="batch", choices=["batch", "frozen"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--roofline-only", action="store_true",
                    help="launch only the roofline kernels (the rocprofv3 --stats profile of this mode is the per-kernel "
                         "average that must agree with roofline.avg_ms)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    dp = DataParallel(backend="nccl")
    assert dp.world == args.gpus or (args.gpus == 1 and dp.world == 1), "launch with torch.distributed.run for --gpus > 1"

    if args.roofline_only:
        r = roofline_leg(args.batch)
        print(json.dumps({k: v["ms"] for k, v in r.items()}))
        return
    log("building engine (batch %d per GPU, %d GPU)" % (args.batch, dp.world))
    model, eng = build_engine(args)
    log("engine built: %d fwd ops, %d bwd ops, %.2f GB device memory" % (
        len(eng.ops_fwd), len(eng.ops_bwd), torch.cuda.memory_allocated() / 1e9))
    dp.broadcast(eng.params)
    dp.broadcast(eng.state)

    def step():
        eng.fwd_bwd()
        scale = dp.allreduce_grads(eng.grads)
        eng.adam(None, scale)

    for i in range(max(args.warmup, 2)):  # >= 2: the first call runs eagerly, the second captures the hipGraph
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    dt = dp.max_over_ranks(time.perf_counter() - t0)
    loss = float(eng.loss[0].item())
    log("timed %d steps: %.1f ms/step" % (args.steps, 1e3 * dt / args.steps))

    if dp.rank == 0:
        imgs = args.batch * dp.world * args.steps
        default_cfg = (args.backbone, args.size, args.head) == ("mobilenetv2", 512, "deeplab")
        metric = "images/sec fwd+bwd, 512x512 MobileNetV2 OS=16, 21 classes" if default_cfg else \
            "images/sec fwd+bwd, %dx%d %s OS=%d head=%s, 21 classes" % (args.size, args.size, args.backbone, args.os, args.head)
        rec = {
            "metric": metric,
            "value": imgs / dt, "unit": "img/s", "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Deeplabv3(backbone='%s', input_shape=(%d,%d,3), classes=21, OS=%d) head=%s: "
                                   "fwd + sparse-xent loss + bwd + Adam, BN %s mode, dropout 0.1"
                                   % (args.backbone, args.size, args.size, args.os, args.head, args.bn_mode),
                       "global_batch": args.batch * dp.world, "per_gpu_batch": args.batch,
                       "parallelism": "dp%d" % dp.world, "hipgraph": eng.graph is not None, "final_loss": loss},
        }
        if not args.no_roofline:
            r = roofline_leg(args.batch)
            log("roofline leg done: " + ", ".join("%s %.3f ms" % (k, v["ms"]) for k, v in r.items()))
            k = r["dw_r4_fwd"]
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, collected separately
            # with tools/dw_only.py and corrected as MI355X_MICROARCH.md prescribes): committed under profiles/
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_dw_r4_b%d_pmc.json" % args.batch)
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                if pj.get("batch") == args.batch:
                    traffic = pj["dw_march_fwd"]["traffic_bytes"]
            rec["roofline"] = {"bound": "hbm", "kernel": "dw_march2_fwd (DepthwiseConv2D 3x3 rate 4, %dx64x64x960)" % args.batch,
                               "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["gbs"] / HBM_PEAK_GBS,
                               "traffic": traffic, "avg_ms": k["ms"], "algorithmic_bytes": k["bytes"]}
            # the kernel family with the largest share of the step is the fp32-MFMA 1x1-conv GEMM (forward shown:
            # Conv2D 1x1 160 -> 960 with BN+ReLU6 on load and BN partial sums in the epilogue)
            kg = r["pw_expand_160_960"]
            rec["roofline_mfma"] = {"bound": "mfma", "kernel": "pw_gemm_stream_kernel (Conv2D 1x1 160->960, M=%d)" % (args.batch * 4096),
                                    "achieved": kg["tflops"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": kg["tflops"] / FP32_PEAK_TFLOPS, "avg_ms": kg["ms"],
                                    "algorithmic_flops": 2.0 * args.batch * 4096 * 160 * 960}
            rec["kernels"] = r
        if dp.world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_leg(args)
            log("cpu baseline done")
        print(json.dumps(rec))
    dp.close()


if __name__ == "__main__":
    main()
