"""Per-image data parallelism: one process per GPU, RCCL all-reduce of the flat gradient arena over xGMI.

Replaces the reference's single-process in-graph `keras.utils.multi_gpu_model` (utils.py:209-211), which
splits the batch across towers and merges on the CPU with no collectives.  Here every rank owns B images
(BatchNorm statistics stay per replica, exactly like the reference's towers), runs the same plan, and ONE
all-reduce(sum) of the flat fp32 gradient arena (8.45 MB MobileNetV2, 164 MB Xception) crosses xGMI per step; the
1/world factor is folded into the Adam kernel (dl3_adam_step grad_scale).  The path has no other exchange step, so
there is no other collective.

Two planes:
  * data plane  — libdl3.so's own RCCL binding (include/dl3.h: dl3_comm_unique_id / _init / _allreduce_f32 /
                  _broadcast_f32 / _destroy), enqueued on the compute stream right behind the backward hipGraph.
                  Bucketing / overlap with backward is deliberately absent: at the benchmarked batch the exchange is
                  0.1-0.3 % of the step (8.45 MB in ~0.15 ms against 58 ms; 164 MB in ~2.5 ms against 800 ms), and
                  a second stream would only add an event round trip.
  * control plane — torch.distributed (gloo, host TCP) carries the 128-byte RCCL id, barriers and the max-over-ranks
                  of the timing.  It never touches device data.
`DL3_DIST_BACKEND=gloo` moves the data plane onto gloo too (gradients staged through host memory): that is how the
CPU tests and the two-processes-on-one-GPU test exercise the N>1 logic (RCCL refuses two ranks on one device).
"""
import ctypes
import os

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, backend=None, device=None, strict=None):
        # strict: a failed RCCL bring-up raises instead of falling back to gloo (None: DL3_DIST_STRICT=1; Model.distribute()
        # passes True — an explicit DL3_DIST_BACKEND=gloo / backend="gloo" is not a fall-back and is always honoured)
        self.strict = (os.environ.get("DL3_DIST_STRICT", "0") == "1") if strict is None else bool(strict)
        if os.environ.get("DL3_DIST_STRICT") == "0":
            self.strict = False
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        backend = os.environ.get("DL3_DIST_BACKEND", backend)
        if backend in (None, "nccl"):  # "nccl" kept as an alias: RCCL is what answers to it on ROCm
            backend = "rccl" if torch.cuda.is_available() else "gloo"
        if backend not in ("rccl", "gloo"):
            raise ValueError("data-parallel backend must be 'rccl' or 'gloo', not %r" % (backend,))
        self.backend = backend
        self.comm = None
        if self.world > 1:
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank % max(torch.cuda.device_count(), 1))
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            if backend == "rccl":
                self._init_rccl()

    def _init_rccl(self):
        from . import capi
        L = capi.lib()
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            raw = ctypes.create_string_buffer(128)
            capi.check(L.dl3_comm_unique_id(raw), "dl3_comm_unique_id")
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        dist.broadcast(idbuf, src=0)  # control plane: 128 bytes over host TCP
        handle = ctypes.c_void_p()
        rc = L.dl3_comm_init(ctypes.byref(handle), bytes(idbuf.numpy().tobytes()), self.rank, self.world)
        # every rank must end up on the same data plane: agree on the outcome over the control plane
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            self.comm = handle
            return
        if rc == 0:
            L.dl3_comm_destroy(handle)
        if self.strict:
            capi.check(rc or 1, "dl3_comm_init (on some rank; strict data plane: no gloo fall-back — "
                       "Model.distribute(strict=False) or DL3_DIST_STRICT=0 allows it)")
        import warnings
        warnings.warn("dl3_comm_init failed on at least one rank (%s): gradients will be exchanged over gloo through "
                      "host memory, NOT over RCCL/xGMI — expect the exchange to dominate small steps"
                      % (L.dl3_last_error().decode() if rc else "this rank was fine"))
        self.backend = "gloo"

    def attach_single_rank_rccl(self):
        """world of one only: bring up an RCCL communicator of ONE rank on the current device, so that the data-parallel
        step (count all-reduce, hipGraph replay, arena all-reduce, Adam with 1/world) runs launch for launch as it does on
        N GPUs.  What a one-GPU box can check of the N-GPU path (tests/test_gpu_parallel.py)."""
        assert self.world == 1 and self.comm is None
        from . import capi
        L = capi.lib()
        raw = ctypes.create_string_buffer(128)
        capi.check(L.dl3_comm_unique_id(raw), "dl3_comm_unique_id")
        handle = ctypes.c_void_p()
        capi.check(L.dl3_comm_init(ctypes.byref(handle), raw.raw, 0, 1), "dl3_comm_init")
        self.comm, self.backend = handle, "rccl"
        return self

    def rccl_ranks(self):
        """number of ranks RCCL itself reports for the data-plane communicator (None when the data plane is not RCCL)"""
        if self.comm is None:
            return None
        from . import capi
        n = ctypes.c_int(0)
        capi.check(capi.lib().dl3_comm_count(self.comm, ctypes.byref(n)), "dl3_comm_count")
        return int(n.value)

    def shard(self, n_global):
        """contiguous image shard [lo, hi) of this rank; a batch that does not divide evenly gives the remainder to the
        last rank — keras.utils.multi_gpu_model's get_slice (utils.py:209-211) does the same with its last tower"""
        per = n_global // self.world
        lo = self.rank * per
        return lo, (n_global if self.rank == self.world - 1 else lo + per)

    def allreduce_grads(self, flat):
        """sum the flat gradient arena over ranks in place; returns the scale the optimizer must apply"""
        if self.comm is not None:   # (also a communicator of one rank: the launch is the same, the sum an identity)
            from . import capi
            capi.call("dl3_comm_allreduce_f32", self.comm, flat.data_ptr(), flat.data_ptr(), flat.numel(),
                      torch.cuda.current_stream().cuda_stream)
        elif self.world > 1:
            if flat.is_cuda:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return 1.0 / self.world

    def broadcast(self, flat, src=0):
        """identical initial weights / Adam state on every rank"""
        if self.comm is not None:
            from . import capi
            capi.call("dl3_comm_broadcast_f32", self.comm, flat.data_ptr(), flat.numel(), src,
                      torch.cuda.current_stream().cuda_stream)
        elif self.world > 1:
            if flat.is_cuda:
                h = flat.cpu()
                dist.broadcast(h, src=src)
                flat.copy_(h)
            else:
                dist.broadcast(flat, src=src)

    def max_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def mean_over_ranks(self, value):
        if self.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item()) / self.world

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def close(self):
        if self.comm is not None:
            from . import capi
            torch.cuda.synchronize()
            capi.lib().dl3_comm_destroy(self.comm)
            self.comm = None
        if self.world > 1 and dist.is_initialized():
            dist.destroy_process_group()
