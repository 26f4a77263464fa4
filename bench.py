"""bench.py — images/sec forward+backward(+Adam) of the DeepLabV3+ MobileNetV2 path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: bench.py re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU); started by torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one resident batch: forward, loss, backward (one replayed hipGraph of
libdl3.so launches), RCCL all-reduce of the flat gradient arena (N>1, dl3_comm_allreduce_f32), Adam.  Inputs are
synthetic and already in HBM when the timed region starts.  Workload = BASELINE.json configs[1]:
Deeplabv3(backbone='mobilenetv2', input_shape=(512,512,3), classes=21, OS=16), fp32, BatchNorm in training
(batch-statistics) mode, dropout on.  `--head subpixel` = configs[2], `--backbone xception --os 8` = configs[3].

`roofline` / `roofline_hbm` are measured IN SITU: after the timed region the same plan is launched eagerly on the same
resident buffers with a HIP event between every two launches; each launch's algorithmic FLOPs / bytes (SURVEY §8d
formulas, from its own arguments) are summed per kernel family and divided by the family's summed device time.
`roofline` is the family with the largest share of the step (the fp32-MFMA 1x1-conv GEMMs), `roofline_hbm` the
dilated depthwise 3x3 convolutions (the north star's "atrous branches").  `cpu_baseline` (N=1): the torch-CPU
restatement of the same step on this box's host cores (kind "port": the reference's Keras/TF path cannot be installed).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # fp32 MFMA = fp32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128,
                    help="images per GPU (8x SegModel.batch_size, utils.py:162; 98 GB of the 288 GB HBM; throughput by "
                         "batch on one MI355X, round 2: 16 -> 945, 32 -> 1042, 64 -> 1108, 128 -> 1140 img/s)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--backbone", default="mobilenetv2")
    ap.add_argument("--os", type=int, default=16, help="output stride (Xception only; MobileNetV2 always runs at 8)")
    ap.add_argument("--head", default="deeplab", choices=["deeplab", "original", "subpixel"])
    ap.add_argument("--bn-mode", default="batch", choices=["batch", "frozen"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches (profiling aid; never a headline number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-split-leg", action="store_true",
                    help="skip the second timed loop with DL3_GEMM_MATH=split (reported beside, never as, the headline)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the compact legs (configs[1] at B=2 / B=16, configs[2], configs[3]) of the default run")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = sweep thread counts up to os.cpu_count()")
    ap.add_argument("--plan-json", default=None, help="write the per-launch in-situ table (op, shape, ms, FLOPs, bytes)")
    return ap.parse_args()


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("spawning %d ranks: %s" % (args.gpus, " ".join(cmd[2:9])))
    env = dict(os.environ)
    if env.get("DL3_DIST_BACKEND") != "gloo":
        env.setdefault("DL3_DIST_STRICT", "1")  # an N>1 number must come from RCCL or not at all (see main)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


def build_engine(args):
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    from dl3_amd.utils import SegModel
    G.clear_session(seed=1)
    shape = (args.size, args.size, 3)
    if args.head == "deeplab":
        model = Deeplabv3(weights=None, input_shape=shape, classes=21, backbone=args.backbone, OS=args.os)
    else:
        model = SegModel(image_size=shape[:2]).create_seg_model(args.head, n=21, backbone=args.backbone)
    kw = {}
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the engine Model.train_on_batch builds under Model.distribute(): ONE loss over the global batch — the shard's
        # count(w != 0) and loss sum ride in the arena all-reduce, Adam finishes the normalisation on the device
        # (Engine.train_step) — the N>1 number times the product path
        kw["external_nnz"] = True
    eng = model._engine(args.batch, True, bn_mode=args.bn_mode, dropout=True, use_graph=not args.no_graph, **kw)
    rng = np.random.default_rng(1000 + int(os.environ.get("RANK", "0")))
    x = rng.integers(0, 256, (args.batch,) + shape).astype(np.float32)
    y = rng.integers(0, 22, (args.batch, args.size * args.size)).astype(np.float32)  # 21 = void
    eng.set_input(x)
    eng.set_targets(y, (y < 21).astype(np.float32))  # the generator's contract: weight 0 on void pixels
    return model, eng


# ---------------------------------------------------------------------------------------------------------------
# in-situ roofline: algorithmic work of every launch of the plan, from its own arguments (include/dl3.h signatures)
# ---------------------------------------------------------------------------------------------------------------
def _work(name, a):
    """(family, shape string, algorithmic FLOPs, algorithmic bytes) of one C-ABI launch; None = not accounted"""
    nz = lambda p: p is not None and p != 0
    if name in ("dl3_pwconv_fwd", "dl3_pwconv_fwd_add"):
        M, K, N = a[9], a[10], a[11]
        return "gemm", "fwd M=%d K=%d N=%d" % (M, K, N), 2.0 * M * K * N, 4.0 * (M * K + M * N + K * N)
    if name == "dl3_pwconv_bwd_data":
        M, K, N = a[22], a[23], a[24]
        by = M * N * (2 if nz(a[2]) else 1) + K * N + M * K * (1 + (1 if nz(a[10]) else 0) + (1 if nz(a[15]) else 0))
        return "gemm", "bwd-data M=%d K=%d N=%d" % (M, K, N), 2.0 * M * K * N, 4.0 * by
    if name in ("dl3_pwconv_bwd_weight", "dl3_pwconv_bwd_weight_dy"):
        M, K, N = a[14], a[15], a[16]
        # (_dy: the launch also writes dY once — the M*N the bwd-data launch of the layer no longer reads twice)
        by = M * K + M * N * (2 if nz(a[7]) else 1) + K * N + (M * N if name.endswith("_dy") else 0)
        return "gemm", "bwd-weight M=%d K=%d N=%d" % (M, K, N), 2.0 * M * K * N, 4.0 * by
    if name == "dl3_pwconv_bwd_fused":
        # both gradients in one pass: x, g (+ yraw), wT in; dx out (+ the addend and, if it is another tensor, stat_x in)
        M, K, N = a[23], a[24], a[25]
        by = M * K + M * N * (2 if nz(a[7]) else 1) + K * N + M * K * (1 + (1 if nz(a[16]) else 0) +
                                                                      (1 if (nz(a[18]) and a[18] != a[0]) else 0))
        return "gemm", "bwd-fused M=%d K=%d N=%d" % (M, K, N), 4.0 * M * K * N, 4.0 * by
    if name == "dl3_dwconv3x3_fwd":
        N, H, W, C, stride, rate, Ho, Wo = a[6], a[7], a[8], a[9], a[10], a[11], a[14], a[15]
        fam = "dw_dilated" if (rate > 1 and stride == 1) else "dw"
        return (fam, "fwd %dx%dx%dx%d s%d r%d" % (N, H, W, C, stride, rate), 2.0 * 9 * N * Ho * Wo * C,
                4.0 * (N * H * W * C + N * Ho * Wo * C + 9 * C))
    if name in ("dl3_dwconv3x3_bwd", "dl3_dwconv3x3_bwd_sx"):
        o = 1 if name.endswith("_sx") else 0   # (_sx: one more operand, stat_x, behind dx_add)
        N, H, W, C, stride, rate, Ho, Wo = a[16 + o], a[17 + o], a[18 + o], a[19 + o], a[20 + o], a[21 + o], a[24 + o], a[25 + o]
        fam = "dw_dilated" if (rate > 1 and stride == 1) else "dw"
        out_e, in_e = N * Ho * Wo * C, N * H * W * C
        by = out_e * (2 if nz(a[1]) else 1) + in_e * (1 + (1 if nz(a[10]) else 0) + (1 if nz(a[11]) else 0) + o) + 18 * C
        return fam, "bwd %dx%dx%dx%d s%d r%d" % (N, H, W, C, stride, rate), 2.0 * 2 * 9 * out_e, 4.0 * by
    if name == "dl3_reduce_partials_batched":
        # every weight-gradient slab fold of the pass in one launch: charged to the GEMM family (no FLOPs), as the fold
        # behind each dl3_pwconv_bwd_weight launch used to be (a few of its entries are depthwise weight gradients)
        return "gemm", "slab folds x%d" % a[1], 0.0, 0.0
    return "other", "", 0.0, 0.0


def insitu_profile(eng, passes=3):
    """per-launch device time of the real plan on the resident batch: eager launches, one HIP event between every two
    launches (on the stream they are launched on), mean over `passes` full steps"""
    st = torch.cuda.current_stream().cuda_stream
    ops = list(eng.ops_fwd) + list(eng.ops_bwd)
    acc = [0.0] * len(ops)
    for _ in range(passes):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        ev[0].record()
        for i, (name, fn, a, _) in enumerate(ops):
            rc = fn(*a, st)
            assert rc == 0, name
            ev[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(ops)):
            acc[i] += ev[i].elapsed_time(ev[i + 1]) / passes
    rows = []
    for (name, fn, a, _), ms in zip(ops, acc):
        fam, shape, fl, by = _work(name, a)
        rows.append(dict(op=name, family=fam, shape=shape, ms=ms, flops=fl, bytes=by))
    return rows


def _pmc_traffic(family, args):
    """(HBM bytes per step, source) of a kernel family from the COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, separate runs, corrected as MI355X_MICROARCH.md prescribes; tools/collect_profiles.sh +
    tools/pmc_family.py).  It is a constant read from profiles/, not a measurement of this run — the line says so in
    `traffic_source` — and only reported for the configuration it was collected on, else (None, None)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        rel = os.path.join("profiles", "%s_%s_b%d_pmc.json" % (rnd, family, args.batch))
        pmc = os.path.join(ROOT, rel)
        if os.path.exists(pmc) and args.head == "deeplab" and args.size == 512:
            pj = json.load(open(pmc))
            if pj.get("batch") == args.batch and pj.get("backbone") == args.backbone:
                return pj.get("traffic_bytes_per_step"), "%s (committed rocprofv3 PMC pass of this configuration, not this run)" % rel
    return None, None


def roofline_blocks(rows, args):
    tot = sum(r["ms"] for r in rows)
    fam = {}
    for r in rows:
        f = fam.setdefault(r["family"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        f["ms"] += r["ms"]
        f["flops"] += r["flops"]
        f["bytes"] += r["bytes"]
        f["launches"] += 1
    out = {"insitu_step_ms": tot,
           "insitu_share": {k: round(v["ms"] / tot, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}
    note = ("in situ: the step's own launches on the resident batch, eager, one HIP event between launches, mean of 3 "
            "steps; achieved = sum of algorithmic %s of the family's launches / sum of their device time")
    g = fam.get("gemm")
    if g and g["ms"] > 0:
        tf = g["flops"] / g["ms"] / 1e9
        out["roofline"] = {
            "bound": "mfma", "kernel": "Conv2D 1x1 GEMM family on v_mfma_f32_32x32x2_f32: pw_gemm_stream_kernel (forward, "
            "bwd-data) + pw_ws2_kernel (short reductions, weight slice resident in LDS) + pw_fwd_ws_kernel (forward of the "
            "HBM-bound early layers) + pw_wgrad_kernel / pw_wgrad_row_kernel (bwd-weight) + pw_bwd_fused2_kernel (both "
            "gradients of the HBM-bound early layers in one pass) + the logits layer's pw_narrowk_kernel / pw_wgrad_narrow_kernel, "
            "%d launches/step" % g["launches"],
            "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
            "traffic": _pmc_traffic("gemm", args)[0], "traffic_source": _pmc_traffic("gemm", args)[1],
            "avg_ms": g["ms"] / g["launches"], "family_ms_per_step": g["ms"], "algorithmic_flops_per_step": g["flops"],
            "algorithmic_bytes_per_step": g["bytes"], "share_of_step": g["ms"] / tot, "note": note % "FLOPs"}
    d = fam.get("dw_dilated")
    if d and d["ms"] > 0:
        gbs = d["bytes"] / d["ms"] / 1e6
        traffic, traffic_src = _pmc_traffic("dw_dilated", args)
        out["roofline_hbm"] = {
            "bound": "hbm", "kernel": "DepthwiseConv2D 3x3 rate>1 family: dw_march2_fwd / dw_march_fwd / dw_march_bwd, "
            "%d launches/step" % d["launches"],
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_ms": d["ms"] / d["launches"], "family_ms_per_step": d["ms"], "algorithmic_bytes_per_step": d["bytes"],
            "share_of_step": d["ms"] / tot, "note": note % "bytes"}
        fw = [r for r in rows if r["family"] == "dw_dilated" and r["shape"].startswith("fwd")]
        if fw:
            b, m = sum(r["bytes"] for r in fw), sum(r["ms"] for r in fw)
            out["roofline_hbm"]["forward_only"] = {"achieved": b / m / 1e6, "frac": b / m / 1e6 / HBM_PEAK_GBS,
                                                   "ms_per_step": m, "launches": len(fw)}
    return out


# ---------------------------------------------------------------------------------------------------------------
def host_cpu_grant():
    """what this process may actually use of the host: the affinity mask, the cgroup CPU quota
    (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1), SMT siblings and the NUMA layout from sysfs"""
    def cpulist(txt):
        out = []
        for part in txt.strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            out += list(range(int(lo), int(hi or lo) + 1))
        return out

    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except Exception:
            quota = None
    nodes = {}
    base = "/sys/devices/system/node"
    try:
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                nodes[int(d[4:])] = [c for c in cpulist(open(os.path.join(base, d, "cpulist")).read()) if c in aff]
    except Exception:
        nodes = {}
    # one hardware thread per physical core of the mask (the first sibling of each core)
    cores = []
    for c in aff:
        try:
            sib = cpulist(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read())
        except Exception:
            sib = [c]
        if c == min(s for s in sib if s in aff or s == c):
            cores.append(c)
    return dict(affinity=aff, quota=quota, nodes={k: v for k, v in nodes.items() if v}, physical=cores)


def cpu_baseline_leg(args):
    """CPU restatement of the same training step (oracle/torch_ref.py: torch/oneDNN on the host cores), SURVEY §8d
    protocol: thread counts swept up to os.cpu_count() (one probe step each), then the median of 10 steps after 2
    warm-ups at the best count, for cfg1's shape (128x128x2, B=2) and cfg2 (512x512x21, B=2).  kind='port': the reference's own
    Keras/TensorFlow CPU path cannot be installed here."""
    from oracle import dl3_oracle as O
    from oracle import torch_ref as T
    grant = host_cpu_grant()
    # the cores actually granted: physical cores of the affinity mask inside ONE NUMA node (the one with most of them: a
    # oneDNN thread pool straddling two sockets scales negatively — that, not oneDNN, was rounds 1-4's "slower at 32 than at
    # 16 threads"), capped by the cgroup quota; the leg pins itself there and restores the mask afterwards
    node_cpus = max(grant["nodes"].values(), key=len) if grant["nodes"] else grant["affinity"]
    pin = [c for c in grant["physical"] if c in node_cpus] or node_cpus
    if grant["quota"]:
        pin = pin[:max(1, int(grant["quota"]))]
    avail = len(pin)
    old_aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if old_aff is not None:
        os.sched_setaffinity(0, pin)
    log("cpu baseline: affinity %d hw threads, cgroup quota %s, %d NUMA node(s) -> pinned to %d physical core(s) of one node" % (
        len(grant["affinity"]), grant["quota"], len(grant["nodes"]) or 1, avail))

    def case(size, classes, B):
        kw = dict(backbone=args.backbone, input_shape=(size, size, 3), classes=classes, OS=args.os)
        params = O.init_params(O.param_shapes(args.backbone, classes), seed=1)
        rng = np.random.default_rng(0)
        x = rng.integers(0, 256, (B, size, size, 3)).astype(np.float32)
        y = rng.integers(0, classes + 1, (B, size, size)).astype(np.float32)
        w = (y < classes).astype(np.float32)
        return lambda: T.train_grads(params, x, y, w, **kw)

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def measure(fn, B, what):
        cands = [args.cpu_threads] if args.cpu_threads > 0 else sorted({t for t in (4, 8, 16, 32, 64, 128, avail) if t <= avail})
        sweep = {}
        for t in cands:
            torch.set_num_threads(t)
            fn()  # warm-up at this count (oneDNN primitive cache, thread pool)
            sweep[t] = B / timed(fn)
            if sweep[t] < 0.5 * max(sweep.values()):
                break  # oversubscribed: larger counts only get slower (and each probe costs seconds)
        best = max(sweep, key=sweep.get)
        torch.set_num_threads(best)
        for _ in range(2):
            fn()
        ts = sorted(timed(fn) for _ in range(10))
        med = 0.5 * (ts[4] + ts[5])
        log("cpu baseline %s: %.2f img/s on %d threads (sweep %s)" % (
            what, B / med, best, {k: round(v, 2) for k, v in sweep.items()}))
        return dict(value=B / med, cores=best, sweep_img_s={str(k): round(v, 3) for k, v in sweep.items()})

    def case_fwd1(size, classes):
        # BASELINE.json configs[0] as stated: Deeplabv3(mobilenetv2, (128,128,3), classes=2, OS=16), single-image
        # forward (inference-mode BatchNorm, moving statistics)
        kw = dict(backbone="mobilenetv2", input_shape=(size, size, 3), classes=classes, OS=16)
        params = O.init_params(O.param_shapes("mobilenetv2", classes), seed=1)
        x = np.random.default_rng(0).integers(0, 256, (1, size, size, 3)).astype(np.float32)
        return lambda: T.infer_logits(params, x, **kw)

    try:
        c2 = measure(case(args.size, 21, 2), 2, "cfg2 %dx%d B=2" % (args.size, args.size))
        cpu_a = cpu_a_leg(args, avail)
        c1 = measure(case_fwd1(128, 2), 1, "cfg1 128x128 single-image forward")
        # cfg1's shape as a training step, B=2 (with one image the image-pooling BatchNorm sees a single value per channel)
        c1t = measure(case(128, 2, 2), 2, "cfg1 128x128 B=2 fwd+bwd")
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    return dict(value=c2["value"], unit="img/s", cores=c2["cores"], kind="port", host_cores_available=avail,
                host={"affinity_hw_threads": len(grant["affinity"]), "cgroup_cpu_quota": grant["quota"],
                      "numa_nodes": {str(k): len(v) for k, v in grant["nodes"].items()},
                      "physical_cores_in_mask": len(grant["physical"]), "pinned_to": "%d physical cores of one NUMA node" % avail},
                sample="median of 10 steps x 2 images %dx%dx21 fwd+bwd after 2 warm-up steps, torch-CPU (oneDNN) restatement "
                       "oracle/torch_ref.py; thread counts swept (4 ... all cores, stopping once a count is half as fast as "
                       "the best): oneDNN's fp32 convolutions of this graph stop scaling at `cores` threads on this host — "
                       "more threads are SLOWER (see thread_sweep_img_s)" % (args.size, args.size),
                thread_sweep_img_s=c2["sweep_img_s"], cpu_a=cpu_a,
                cfg1={"value": c1["value"], "unit": "img/s", "cores": c1["cores"],
                      "sample": "BASELINE.json configs[0]: median of 10 single-image forwards 128x128x3 -> 2 classes "
                                "(inference BatchNorm) after 2 warm-ups", "thread_sweep_img_s": c1["sweep_img_s"]},
                cfg1_train_b2={"value": c1t["value"], "unit": "img/s", "cores": c1t["cores"],
                               "sample": "median of 10 steps x 2 images 128x128x2 fwd+bwd",
                               "thread_sweep_img_s": c1t["sweep_img_s"]})


def cpu_a_leg(args, avail):
    """CPU-A: the repo's own C / OpenMP restatement of the step — oracle/dl3_oracle.py's graph on the
    operators of oracle/c/dl3_ops.c, float32 — cfg2 at B=2: one probe step per thread count, then the median of 3
    steps after 1 warm-up at the best count (a step takes seconds: bounded sample)."""
    from oracle import c_backend as CB
    from oracle import dl3_oracle as O
    kw = dict(backbone=args.backbone, input_shape=(args.size, args.size, 3), classes=21, OS=args.os)
    params = O.init_params(O.param_shapes(args.backbone, 21), seed=1)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (2, args.size, args.size, 3)).astype(np.float32)
    y = rng.integers(0, 22, (2, args.size * args.size)).astype(np.float32)
    w = (y < 21).astype(np.float32)

    def step():
        t0 = time.perf_counter()
        O.train_grads(params, x, y, w, **kw)
        return time.perf_counter() - t0

    sweep = {}
    with CB.installed():
        for t in sorted({t for t in (8, 16, 32, 64, 128) if t <= avail}):
            CB.set_threads(t)
            sweep[t] = 2 / step()
            if sweep[t] < 0.6 * max(sweep.values()):
                break
        best = max(sweep, key=sweep.get)
        CB.set_threads(best)
        step()
        ts = sorted(step() for _ in range(3))
    log("cpu-a (C / OpenMP restatement) cfg2 B=2: %.2f img/s on %d threads (sweep %s)" % (
        2 / ts[1], best, {k: round(v, 2) for k, v in sweep.items()}))
    return dict(value=2 / ts[1], unit="img/s", cores=best, kind="port",
                sample="median of 3 steps x 2 images %dx%dx21 fwd+bwd after 1 warm-up, numpy graph of oracle/dl3_oracle.py on "
                       "the C / OpenMP operators of oracle/c/dl3_ops.c (float32)" % (args.size, args.size),
                thread_sweep_img_s={str(k): round(v, 3) for k, v in sweep.items()})


def capi_math():
    from dl3_amd import capi
    return capi.get_gemm_math()


def split_math_leg(args):
    """The same K steps with the 1x1-conv GEMMs in split math (dl3_set_gemm_math(DL3_MATH_SPLIT): every fp32 operand
    cut EXACTLY into three bf16 pieces, six of the nine piece products on v_mfma_f32_32x32x16_bf16, fp32 accumulate;
    csrc/pwgemm.hip split3).  Reported BESIDE the headline, which stays on the f32 MFMA."""
    import gc
    from dl3_amd import graph as G
    gc.collect()
    G.clear_session()
    torch.cuda.empty_cache()
    from dl3_amd import capi
    capi.set_gemm_math("split")
    try:
        model, eng = build_engine(args)
        for _ in range(max(args.warmup, 2)):
            eng.fwd_bwd()
            eng.adam(None, 1.0)
        torch.cuda.synchronize()
        if not args.no_graph and eng.graph is None:
            return {"error": "hipGraph capture failed"}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.fwd_bwd()
            eng.adam(None, 1.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss = eng.loss_value()
    finally:
        capi.set_gemm_math(None)
    log("split-math leg: %.1f ms/step" % (1e3 * dt / args.steps))
    return {"value": args.batch * args.steps / dt, "unit": "img/s", "ms_per_step": 1e3 * dt / args.steps, "final_loss": loss,
            "what": "same workload, same K steps; the 1x1-conv GEMMs (forward, bwd-data, bwd-weight) computed as 6 bf16 MFMAs "
                    "on exact 3-way bf16 splits of the fp32 operands (fp32 accumulate); everything else unchanged",
            "accuracy": "max |err| / sum|a||b| against float64 at M=65536 K=960 N=160: split 3.3e-7, v_mfma_f32 3.7e-7 "
                        "(tests/test_gpu_ops.py::test_split_math_error); full-size parity in this mode: tests/test_gpu_fullsize.py "
                        "::test_cfg2_mnv2_512_train_step_split_math, ::test_cfg3_subpixel_mnv2_512_train_step_split_math, "
                        "::test_cfg4_xception_os8_256_train_step_split_math (same bars as the f32 tests)"}


def fed_run(eng, steps, classes=21, step=None, dp=None):
    """The same resident step FED from the host (utils.py:360-402 + :231-241 are what it replaces): every
    step a NEW batch — uint8 images + uint8 label maps, three distinct pinned host batches in rotation — crosses PCIe on a
    copy stream while the previous step runs (feed.BatchFeeder: two device slots), is widened into the engine's input and
    turned into (Y, SW) by dl3_prepare_targets on the device.  Timed like the resident loop, on the same engine.
    N > 1 (round 6): every rank feeds ITS shard (its own host batches, its own copy stream) and runs the
    data-parallel step handed in — captured step, ONE arena all-reduce, Adam finishing the scale on the device —; barrier on both
    sides, max over ranks, whole-job images per second."""
    from dl3_amd.feed import BatchFeeder
    B = eng.B
    H, W = eng.xbuf.H, eng.xbuf.W
    rng = np.random.default_rng(77 + (dp.rank if dp is not None else 0))
    world = dp.world if dp is not None else 1
    host = []
    for _ in range(3):
        img = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).pin_memory()
        lab = rng.integers(0, classes + 1, (B, H * W), dtype=np.uint8)
        lab[lab == classes] = 255                     # void as cv2 delivers it
        host.append((img, torch.from_numpy(lab).pin_memory()))
    fd = BatchFeeder(eng, classes, np.uint8)

    if step is None:
        def step():
            eng.fwd_bwd()
            eng.adam(None, 1.0)

    fd.run((host[i % 3] for i in range(3)), step)      # warm-up: pinned registration, copy stream, both slots
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    t0 = time.perf_counter()
    n = fd.run((host[i % 3] for i in range(steps)), step)
    torch.cuda.synchronize()
    if dp is not None:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dt = dp.max_over_ranks(dt)
    assert n == steps
    # the link on its own: one image batch, H2D, alone on the copy stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(fd.copy_stream):
        e0.record(fd.copy_stream)
        fd.dx[0].copy_(host[0][0].reshape(-1), non_blocking=True)
        e1.record(fd.copy_stream)
    torch.cuda.synchronize()
    raw = fd.nx / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return {"value": B * world * steps / dt, "unit": "img/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "host_bytes_per_step": fd.bytes_per_batch, "pcie_gb_s_sustained_by_the_loop": fd.bytes_per_batch * steps / dt / 1e9,
            "pcie_gb_s_one_copy_alone": raw, "final_loss": eng.loss_value()}


def _free_engines():
    import gc
    from dl3_amd import graph as G
    gc.collect()
    G.clear_session()
    torch.cuda.empty_cache()


def compact_leg(args, **over):
    """One more configuration of BASELINE.json timed the same way (resident synthetic batch, hipGraph replay + Adam,
    barrier-free single GPU), compactly: 3 warm-up steps, then as many steps as fill about a second (5..200), plus ONE
    in-situ pass for the family fractions.  The headline engine has been freed before."""
    a = argparse.Namespace(**vars(args))
    for k, v in over.items():
        if k != "fed":
            setattr(a, k, v)
    _free_engines()
    model, eng = build_engine(a)

    def step():
        eng.fwd_bwd()
        eng.adam(None, 1.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if not a.no_graph and eng.graph is None:
        return {"error": "hipGraph capture failed"}
    t0 = time.perf_counter()
    step()
    step()
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 2
    steps = int(min(200, max(5, round(1.0 / max(est, 1e-4)))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"value": a.batch * steps / dt, "unit": "img/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "batch": a.batch,
           "workload": "%s %dx%d OS=%d head=%s B=%d" % (a.backbone, a.size, a.size, a.os, a.head, a.batch),
           "launches_per_step": len(eng.ops_fwd) + len(eng.ops_bwd) + 1, "device_gb": round(torch.cuda.memory_allocated() / 1e9, 2)}
    if over.get("fed"):
        out["fed"] = fed_run(eng, steps)
        out["fed"]["vs_resident"] = round(out["fed"]["value"] / out["value"], 4)
        log("   fed from the host: %.1f img/s = %.3f x resident" % (out["fed"]["value"], out["fed"]["vs_resident"]))
    rb = {} if a.no_roofline else roofline_blocks(insitu_profile(eng, passes=1), a)
    if "roofline" in rb:
        out["gemm_frac_of_fp32_mfma_peak"] = round(rb["roofline"]["frac"], 4)
        out["gemm_share_of_step"] = round(rb["roofline"]["share_of_step"], 4)
    if "roofline_hbm" in rb:
        out["atrous_dw_frac_of_hbm_peak"] = round(rb["roofline_hbm"]["frac"], 4)
        out["atrous_dw_share_of_step"] = round(rb["roofline_hbm"]["share_of_step"], 4)
    log("leg %s: %.1f img/s (%.2f ms/step, %d steps)" % (out["workload"], out["value"], out["ms_per_step"], steps))
    del step
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args)
    import dl3_amd  # noqa: F401
    from dl3_amd.parallel import DataParallel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if world > ndev and os.environ.get("DL3_DIST_BACKEND") != "gloo":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s); RCCL needs one GPU per rank (DL3_DIST_BACKEND=gloo "
                         "oversubscribes a GPU for functional tests only)" % (world, ndev))
    torch.cuda.set_device(local % ndev)
    if world > 1 and os.environ.get("DL3_DIST_BACKEND") != "gloo":
        # an N>1 line must never silently be a host-staged gloo number: a failing ncclCommInitRank aborts the run unless
        # the gloo data plane was asked for explicitly (functional tests on a one-GPU box)
        os.environ.setdefault("DL3_DIST_STRICT", "1")
    dp = DataParallel()
    log("building engine (batch %d per GPU, %d GPU, data plane %s)" % (args.batch, dp.world, dp.backend if dp.world > 1 else "-"))
    model, eng = build_engine(args)
    log("engine built: %d fwd ops, %d bwd ops, %.2f GB device memory" % (
        len(eng.ops_fwd), len(eng.ops_bwd), torch.cuda.memory_allocated() / 1e9))
    dp.broadcast(eng.params)
    dp.broadcast(eng.state)

    def step():
        # (N > 1: nothing in front of the replayed hipGraph, ONE all-reduce behind it — gradients, the shard's
        # count(w != 0) and loss sum in one arena —, Adam with the normalisation finished on the device; no host sync)
        eng.fwd_bwd()
        scale = dp.allreduce_grads(eng.grads)
        eng.adam(None, scale, norm=eng.external_nnz)

    def allreduce_ms(reps=10):
        """the gradient exchange on its own (same arena, same stream, same communicator), HIP events around `reps`
        back-to-back all-reduces after a barrier; max over ranks.  Timed AFTER the timed region and on a scratch copy."""
        if dp.world == 1:
            return None
        scratch = eng.grads.clone()
        dp.allreduce_grads(scratch)
        torch.cuda.synchronize()
        dp.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dp.allreduce_grads(scratch)
        e1.record()
        torch.cuda.synchronize()
        return dp.max_over_ranks(e0.elapsed_time(e1) / reps)

    for i in range(max(args.warmup, 2)):  # >= 2: the first call runs eagerly, the second captures the hipGraph
        step()
        torch.cuda.synchronize()
        log("warm-up step %d done" % i)
    if not args.no_graph and eng.graph is None:
        raise SystemExit("bench.py: hipGraph capture failed — refusing to report eager-launch numbers as the headline")
    dp.barrier()
    torch.cuda.synchronize()
    # no host round trip inside the timed loop: torch raises on any implicit device synchronisation
    # (.item(), .cpu(), a blocking copy) while the guard is on.  Not on the gloo plane: its exchange is host-staged.
    guard = None
    if dp.world == 1 or dp.comm is not None:
        try:
            torch.cuda.set_sync_debug_mode("error")
            guard = 0
        except Exception:   # pragma: no cover - depends on the runtime
            guard = None
    t0 = time.perf_counter()
    try:
        for _ in range(args.steps):
            step()
    finally:
        if guard is not None:
            torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    dp.barrier()
    dt = dp.max_over_ranks(time.perf_counter() - t0)
    loss = eng.loss_value()
    log("timed %d steps: %.1f ms/step" % (args.steps, 1e3 * dt / args.steps))
    ar_ms = allreduce_ms()
    fed = None
    if not args.no_legs and (dp.world > 1 or not eng.external_nnz):
        # (N > 1: every rank feeds its own shard into the data-parallel step; N = 1: the plain resident step)
        try:
            fed = fed_run(eng, args.steps, step=step if dp.world > 1 else None, dp=dp if dp.world > 1 else None)
            fed["vs_resident"] = round(fed["value"] / (args.batch * dp.world * args.steps / dt), 4)
            log("fed from the host: %.1f img/s = %.3f x resident" % (fed["value"], fed["vs_resident"]))
        except Exception as e:   # a side leg must never cost the headline line (N > 1 has only ever run on two gloo ranks here)
            if dp.world == 1:
                raise
            fed = {"error": "%s: %s" % (type(e).__name__, e)}
            log("fed leg failed: %s" % fed["error"])

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
                       "parallelism": "dp%d" % dp.world, "hipgraph": eng.graph is not None, "final_loss": loss,
                       "launches_per_step": len(eng.ops_fwd) + len(eng.ops_bwd) + 1,
                       "matrix_math": capi_math(),
                       "gradient_exchange": ("dl3_comm_allreduce_f32 (RCCL), %.2f MB" % (eng.n_param * 4 / 1e6))
                       if dp.comm is not None else ("gloo (host staged)" if dp.world > 1 else None),
                       "rccl_ranks": dp.rccl_ranks(), "dist_strict": os.environ.get("DL3_DIST_STRICT", "0") == "1",
                       "backward_fork": bool(eng.fork and eng._side),
                       # 0: asserted — the timed loop ran under torch.cuda.set_sync_debug_mode("error"); None: guard unavailable
                       "host_syncs_in_timed_loop": guard},
        }
        if ar_ms is not None:
            rec["allreduce_ms"] = ar_ms  # one exchange of the %d-float arena on its own, max over ranks (part of ms_per_step)
        if fed is not None:
            rec["fed"] = fed   # the same steps fed from pinned host memory (never `value`: that is the resident rate)
        if not args.no_roofline:
            rows = insitu_profile(eng)
            log("in-situ profile done: %.2f ms for %d launches" % (sum(r["ms"] for r in rows), len(rows)))
            rec.update(roofline_blocks(rows, args))
            if args.plan_json:
                with open(args.plan_json, "w") as f:
                    json.dump({"batch": args.batch, "backbone": args.backbone, "head": args.head, "os": args.os,
                               "size": args.size, "rows": rows}, f, indent=0)
        rec["config"]["deferred_fold_workspaces_mb"] = round(eng.own_fold_ws_bytes / 2 ** 20, 1)
        default_run = default_cfg and args.batch == 128 and args.bn_mode == "batch"
        if dp.world == 1 and not args.no_split_leg and rec["config"]["matrix_math"] == "f32":
            del step
            eng = model = None
            rec["split_math"] = split_math_leg(args)
        if dp.world == 1 and default_run and not args.no_legs:
            # what else BASELINE.json lists, timed by the same run: the reference's own batch sizes for
            # configs[1] (notebook: 2, SegModel default: 16, utils.py:162), configs[2] (Subpixel head) and configs[3]
            # (Xception OS=8).  `value` above is untouched by these.
            try:
                del step
            except NameError:
                pass
            eng = model = None
            rec["by_batch"] = {str(b): compact_leg(args, batch=b, fed=(b != 32)) for b in (2, 16, 32)}
            rec["configs"] = {
                "cfg3_subpixel_b128": compact_leg(args, head="subpixel"),
                "cfg4_xception_os8_b16": compact_leg(args, backbone="xception", os=8, batch=16),
            }
            _free_engines()
        if dp.world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_leg(args)
            log("cpu baseline done")
        print(json.dumps(rec), flush=True)
    dp.barrier()  # rank 0's in-situ pass is local work: the others wait here, then every rank tears its communicator down
    dp.close()


if __name__ == "__main__":
    main()
