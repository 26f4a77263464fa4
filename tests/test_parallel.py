"""world_size-2 gloo test of the data-parallel step logic (parallel.py): image shards per rank, one all-reduce
(sum) of the flat gradient arena, 1/world folded into the optimizer — checked against the single-process mean of
the per-shard gradients computed by the CPU oracle."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import dl3_oracle as O

KW = dict(backbone="mobilenetv2", input_shape=(32, 32, 3), classes=3)


def _data():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (4, 32, 32, 3)).astype(np.float32)
    labels = rng.integers(0, 4, (4, 32, 32)).astype(np.float32)
    return x, labels, (labels < 3).astype(np.float32)


def _flat(grads, names):
    return np.concatenate([grads[n].reshape(-1) for n in names]).astype(np.float32)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import dl3_amd  # noqa: F401
    from dl3_amd.parallel import DataParallel
    dp = DataParallel(backend="gloo")
    x, labels, w = _data()
    lo, hi = dp.shard(x.shape[0])
    params = O.init_params(O.param_shapes("mobilenetv2", 3), seed=1 + rank)  # deliberately different per rank
    names = sorted(n for n in params if "/moving_" not in n)
    flat_p = torch.from_numpy(_flat(params, names))
    dp.broadcast(flat_p, src=0)  # identical initial weights on every rank
    off = 0
    for n in names:
        params[n] = flat_p[off:off + params[n].size].numpy().reshape(params[n].shape).copy()
        off += params[n].size
    _, grads, _, _ = O.train_grads(params, x[lo:hi], labels[lo:hi], w[lo:hi], **KW)
    flat_g = torch.from_numpy(_flat(grads, names))
    scale = dp.allreduce_grads(flat_g)
    t = dp.max_over_ranks(1.0 + rank)
    dp.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), g=flat_g.numpy() * scale, lo=lo, hi=hi, tmax=t,
             p0=flat_p.numpy()[:64])
    dp.close()


def test_two_rank_gradient_allreduce(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in (0, 1)]
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 4)
    assert np.array_equal(r0["g"], r1["g"]) and np.array_equal(r0["p0"], r1["p0"])
    assert float(r0["tmax"]) == 2.0 and float(r1["tmax"]) == 2.0
    x, labels, w = _data()
    params = O.init_params(O.param_shapes("mobilenetv2", 3), seed=1)
    names = sorted(n for n in params if "/moving_" not in n)
    gs = [_flat(O.train_grads(params, x[a:b], labels[a:b], w[a:b], **KW)[1], names) for a, b in ((0, 2), (2, 4))]
    want = (gs[0] + gs[1]) / 2
    assert np.allclose(r0["g"], want, rtol=1e-5, atol=1e-6 * np.abs(want).max())


class _StubLib:
    """stands in for libdl3.so's RCCL binding: the communicator comes up on rank 0 and fails on rank 1"""
    def __init__(self, rank):
        self.rank, self.destroyed = rank, 0

    def dl3_comm_unique_id(self, raw):
        return 0

    def dl3_comm_init(self, handle_ref, idbytes, rank, world):
        return 0 if rank == 0 else 5

    def dl3_comm_destroy(self, handle):
        self.destroyed += 1
        return 0

    def dl3_last_error(self):
        return b"stub: no communicator on this rank"


def _worker_fallback(rank, world, port, out_dir):
    import warnings
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("DL3_DIST_BACKEND", None)
    import dl3_amd  # noqa: F401
    from dl3_amd import capi
    from dl3_amd.parallel import DataParallel
    stub = _StubLib(rank)
    capi.lib = lambda: stub
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        dp = DataParallel(backend="rccl")
    g = torch.full((8,), float(rank + 1))
    scale = dp.allreduce_grads(g)
    np.savez(os.path.join(out_dir, "fb%d.npz" % rank), backend=dp.backend, comm=dp.comm is None, g=g.numpy() * scale,
             warned=any("NOT over RCCL" in str(w.message) for w in rec), destroyed=stub.destroyed)
    dp.close()


def test_ranks_agree_when_the_rccl_communicator_fails_on_one_of_them(tmp_path):
    """parallel.py _init_rccl: a rank whose communicator came up must not run the RCCL plane alone — both fall back to
    gloo (with a warning), the healthy rank destroys its communicator, and the exchange still works"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_fallback, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(os.path.join(str(tmp_path), "fb%d.npz" % r)) for r in (0, 1)]
    for r in (r0, r1):
        assert str(r["backend"]) == "gloo" and bool(r["comm"]) and bool(r["warned"])
        assert np.allclose(r["g"], 1.5)
    assert int(r0["destroyed"]) == 1 and int(r1["destroyed"]) == 0


def _worker_strict(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("DL3_DIST_BACKEND", None)
    os.environ.pop("DL3_DIST_STRICT", None)
    import dl3_amd  # noqa: F401
    from dl3_amd import capi
    from dl3_amd.parallel import DataParallel
    stub = _StubLib(rank)
    capi.lib = lambda: stub
    raised = ""
    try:
        DataParallel(backend="rccl", strict=True)   # what Model.distribute() builds by default
    except capi.DL3Error as e:
        raised = str(e)
    np.savez(os.path.join(out_dir, "st%d.npz" % rank), raised=raised, destroyed=stub.destroyed)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def test_strict_data_plane_raises_on_every_rank_instead_of_falling_back(tmp_path):
    """Model.distribute() (strict=True, round 6): when the RCCL communicator fails on ANY rank, EVERY rank raises — no
    silent gloo exchange through host memory — and the rank whose communicator did come up destroys it first"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_strict, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(os.path.join(str(tmp_path), "st%d.npz" % r)) for r in (0, 1)]
    for r in (r0, r1):
        assert "dl3_comm_init" in str(r["raised"]) and "strict" in str(r["raised"])
    assert int(r0["destroyed"]) == 1 and int(r1["destroyed"]) == 0


def _ragged():
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (5, 32, 32, 3)).astype(np.float32)
    labels = rng.integers(0, 3, (5, 32, 32)).astype(np.float32)
    labels[:2][rng.uniform(size=labels[:2].shape) < 0.7] = 3   # rank 0's shard is mostly void
    w = ((labels < 3) * rng.uniform(0.5, 2.0, labels.shape)).astype(np.float32)
    return x, labels, w


def _worker_global_nnz(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import dl3_amd  # noqa: F401
    from dl3_amd.parallel import DataParallel
    dp = DataParallel(backend="gloo")
    x, labels, w = _ragged()
    lo, hi = dp.shard(x.shape[0])
    params = O.init_params(O.param_shapes("mobilenetv2", 3), seed=1)
    names = sorted(n for n in params if "/moving_" not in n)
    loss, grads, _, _ = O.train_grads(params, x[lo:hi], labels[lo:hi], w[lo:hi], bn_frozen=True, **KW)
    # what Engine.train_step does with external_nnz (round 5): every rank differentiates sum_shard(l*w) / c0 with a fixed
    # c0 (the oracle normalised by the shard's own count: rescale), its count(w != 0) and loss sum ride behind the
    # gradients through ONE all-reduce, and the optimizer multiplies by c0 / count_all
    local = float((w[lo:hi] != 0).sum())
    c0 = float(x.shape[1] * x.shape[2])
    g = _flat(grads, names)
    arena = torch.from_numpy(np.concatenate([g * np.float32(local / c0), np.float32([local, loss * local / c0])]))
    dp.allreduce_grads(arena)
    a = arena.numpy()
    n = g.size
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), g=a[:n] * np.float32(c0 / a[n]), lo=lo, hi=hi,
             loss=float(a[n + 1]) * c0 / float(a[n]), count=float(a[n]))
    dp.close()


def test_global_loss_normalisation_over_a_ragged_batch(tmp_path):
    """parallel.DataParallel.shard / allreduce_grads + the rule of Engine.train_step(external_nnz): 5 images over 2
    ranks (2 + 3, the remainder on the last rank like multi_gpu_model's last tower), unequal void fractions; with
    frozen BatchNorm the exchanged gradient and the reported loss equal the single-process step on all 5 images."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_global_nnz, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [np.load(os.path.join(str(tmp_path), "n%d.npz" % r)) for r in (0, 1)]
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 5)
    assert np.array_equal(r0["g"], r1["g"])
    x, labels, w = _ragged()
    params = O.init_params(O.param_shapes("mobilenetv2", 3), seed=1)
    names = sorted(n for n in params if "/moving_" not in n)
    loss, grads, _, _ = O.train_grads(params, x, labels, w, bn_frozen=True, **KW)
    want = _flat(grads, names)
    assert np.allclose(r0["g"], want, rtol=1e-4, atol=1e-6 * np.abs(want).max())
    assert abs(float(r0["loss"]) - loss) < 1e-6 * abs(loss) and float(r0["loss"]) == float(r1["loss"])
    assert float(r0["count"]) == float((w != 0).sum())
