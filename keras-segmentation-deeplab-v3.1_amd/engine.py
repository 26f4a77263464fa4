"""Executor: lowers a graph.Model into a static plan of libdl3.so launches (forward, loss, backward,
Adam) over HBM-resident buffers, and replays it — eagerly or as one captured hipGraph.

Design (MI355X-first):
  * one process per GPU; torch is used only for device memory, streams, graph capture and RCCL.
  * NHWC fp32.  A conv output is written ONCE as the raw pre-BatchNorm tensor; BatchNorm (+ReLU/ReLU6)
    of a producer is a per-channel (scale, shift, act) "view" applied on load by its consumers, so
    the 54 BN + 37 activation passes of the reference graph cost no HBM traffic.
  * BN batch statistics / BN-backward sums / weight gradients are deterministic partial reductions
    written by the producing kernel's epilogue and folded by tiny finalize kernels.
  * Concatenate is zero-copy: producers write channel slices of one buffer (ld = total channels).
  * gradients w.r.t. a buffer are accumulated by its consumers' bwd-data epilogues (mask -> add ->
    BN-backward partial sums), residual Adds are gradient aliases.
Reference call stack being replaced: Keras Model.predict / train_on_batch -> TF Session.run.
"""
import math
import os

import numpy as np
import torch

from . import capi
from .capi import ACT_NONE, ACT_RELU, ACT_RELU6, IMPL_AUTO, ptr

_ACT = {"relu": ACT_RELU, "relu6": ACT_RELU6}
V_SCALE, V_SHIFT, V_MEAN, V_INVSTD, V_CA, V_CB, V_CC, V_NEGMEAN = range(8)


def _unbias_keras224(n, eps):
    """Keras 2.2.4 BatchNormalization.call on the TF 1.13 backend (the reference's environment):
    tf.nn.fused_batch_norm hands back the Bessel-corrected batch variance (n/(n-1)) and the layer multiplies it by
    sample_size / (sample_size - (1.0 + epsilon)) once more before the moving-average update."""
    return (n / (n - 1.0) if n > 1 else 1.0) * (n / (n - (1.0 + eps)) if n > 1.0 + eps else 1.0)


# moving_variance update factor on the biased batch variance (dl3_bn_finalize var_unbias), as a function of (n, eps)
BN_VARIANCE = {
    "keras224": _unbias_keras224,
    "bessel": lambda n, eps: n / (n - 1.0) if n > 1 else 1.0,   # tf.keras / plain FusedBatchNorm
    "biased": lambda n, eps: 1.0,
}


# tf.nn.fused_batch_norm (tensorflow/python/ops/nn_impl.py, TF 1.13) raises every epsilon to at least 1.001e-5 before the
# kernel sees it, and Keras 2.2.4 sends every 4-D BatchNormalization through it in both phases (training:
# K.normalize_batch_in_training -> _fused_normalize_batch_in_training; inference: K.batch_normalization ->
# fused_batch_norm(is_training=False)).  The 1e-5 layers of the ASPP / decoder (deeplabv3p.py:379,386,393-399,408,422-423,
# 427-429) therefore normalise with 1.001e-5.  The engine applies the floor where the Python wrapper does — in front of the
# kernels (dl3_bn_finalize / _direct / _frozen / _frozen_centered take the epsilon they are given); the moving-variance
# factor of BatchNormalization.call keeps the layer's own epsilon.
FUSED_BN_MIN_EPSILON = 1.001e-5


def fused_bn_epsilon(eps):
    eps = float(eps)
    return eps if eps > FUSED_BN_MIN_EPSILON else FUSED_BN_MIN_EPSILON


# BatchNorm over at most this many rows takes its batch statistics straight from the tensor (dl3_bn_finalize_direct)
SMALL_BN_ROWS = 4096


def _same_pads(size, k, stride, rate):
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) * rate + 1 - size, 0)
    return out, total // 2


# ------------------------------------------------------------------ physical channel counts
def stored_channels(c):
    """channels a tensor is STORED with.  A pixel row that is not a whole number of 128-byte lines makes every
    32-channel slab of the depthwise kernels straddle two lines: measured on MI355X, 16x64x64xC rate 2, forward /
    backward: C=728 4.03 / 3.26 TB/s, C=736 6.44 / 4.44 TB/s (tools/dw_rates.py align).  Xception's 728-channel tensors
    (entry_flow_block3 .. exit_flow_block1, 50+ depthwise layers) are therefore stored 736 wide: the 8 extra channels
    carry zero weights, zero gamma / beta, and stay exactly zero forward and backward.  Only when the padding costs
    <= 2 % traffic (144 -> 160 would cost 11 % and gains nothing net)."""
    p = (c + 31) // 32 * 32
    return p if (p - c) <= 0.02 * c else c


def plan_channels(model):
    """{id(tensor): physical channel count}: a Conv2D may widen its output (zero columns), every other layer
    propagates its input's width; producers that write slices of a Concatenate keep their logical width so that the
    slices stay adjacent"""
    nopad = set()
    for l in model._topo:
        if l.kind == "Concatenate":
            for t in l.inbound:
                p = t.layer
                while p.kind not in ("Conv2D", "DepthwiseConv2D", "Add", "InputLayer", "Subpixel", "Concatenate"):
                    p = p.inbound[0].layer
                nopad.add(id(p))
    phys = {}
    for l in model._topo:
        if l is model.input.layer:
            phys[id(l.output)] = l.output.shape[-1]
            continue
        cin = [phys[id(t)] for t in l.inbound]
        if l.kind == "Conv2D":
            f = l.cfg["filters"]
            # only 1x1 convolutions widen their output: the dense 3x3 lowering (Conv3Unit) sizes its kernel and buffers
            # from the logical filter count
            c = f if (id(l) in nopad or l.cfg["k"] != 1) else stored_channels(f)
        elif l.kind == "Subpixel":
            c = l.cfg["out_filters"]
        elif l.kind == "Concatenate":
            c = sum(cin)
        elif l.kind == "Add":
            if len(set(cin)) != 1:
                raise NotImplementedError("Add of tensors stored with different channel counts (%s)" % l.name)
            c = cin[0]
        else:  # same logical width as the input: same stored width; otherwise (none on the path) the logical one
            c = cin[0] if l.inbound[0].shape[-1] == l.output.shape[-1] else l.output.shape[-1]
        phys[id(l.output)] = c
    return phys


def device_shape(phys, l, name, hshape):
    """device shape of a weight: channel dimensions follow the stored width of the layer's input / output"""
    cin = phys[id(l.inbound[0])] if l.inbound else None
    cout = phys[id(l.output)]
    if name.endswith("/depthwise_kernel:0"):
        return (hshape[0], hshape[1], cin, 1)
    if name.endswith("/kernel:0"):
        if l.kind == "Subpixel":
            return (hshape[0], hshape[1], cin, hshape[3])
        return (hshape[0], hshape[1], cin, cout)
    if l.kind == "Subpixel":
        return tuple(hshape)
    return (cout,)  # bias, gamma, beta, moving statistics


class Buf:
    """One device tensor [N*H*W, ld] + its per-channel vectors + gradient accounting."""

    def __init__(self, eng, N, H, W, ld, name, requires_grad=True):
        self.eng, self.N, self.H, self.W, self.ld, self.name = eng, N, H, W, ld, name
        self.M = N * H * W
        self.t = eng.empty(self.M * ld)
        v = torch.zeros(8, ld, dtype=torch.float32)
        v[V_SCALE] = 1.0
        v[V_CA] = 1.0
        self.vec = v.to(eng.device)
        self.bns = []          # (bn_layer, c_off, C)
        self.requires_grad = requires_grad
        self.expected = 0      # gradient contributions to come (consumer units)
        self.done = 0
        self.grad = None       # tensor [M, ld]
        self.addend = None     # pending pass-through gradient (tensor [M, ld])
        self.nonneg = False
        self.centred = set()   # channel offsets of slices stored as y - moving_mean (PwUnit.set_output_offset)

    def vptr(self, row, off=0):
        return self.vec.data_ptr() + 4 * (row * self.ld + off)


class View:
    """A Keras tensor as the engine sees it: channel slice of a Buf, read as act(scale*x+shift)."""

    def __init__(self, buf, off, C, act=ACT_NONE, aff=False, pad=None, shape=None):
        self.buf, self.off, self.C, self.act, self.aff, self.pad = buf, off, C, act, aff, pad
        self.shape = shape or (buf.N, buf.H, buf.W, C)

    def derive(self, **kw):
        d = dict(buf=self.buf, off=self.off, C=self.C, act=self.act, aff=self.aff, pad=self.pad, shape=self.shape)
        d.update(kw)
        return View(**d)

    @property
    def ld(self):
        return self.buf.ld

    def p(self):
        return self.buf.t.data_ptr() + 4 * self.off

    def scale(self):
        return self.buf.vptr(V_SCALE, self.off) if self.aff else None

    def shift(self):
        return self.buf.vptr(V_SHIFT, self.off) if self.aff else None

    def xform(self):
        return (self.scale(), self.shift(), self.act)


class Engine:
    def __init__(self, model, batch, training, bn_mode="batch", dropout=True, seed=2, device=None, use_graph=True,
                 dw_impl=IMPL_AUTO, rank=None, bn_variance="keras224", external_nnz=False, opt_trainable=None, bn_update=None):
        if not torch.cuda.is_available():
            raise capi.DL3Error("the dl3 engine needs a GPU (HIP device); there is no CPU fallback")
        self.lib = capi.lib()
        self.model, self.B, self.training = model, int(batch), bool(training)
        # {layer name: bool}: whose weights the optimizer updates (Keras: collected at compile()) / whose update ops run
        # (Keras: collected when the train function is built); None: the live `layer.trainable` (graph.Model._training_flags)
        self._opt_trainable = opt_trainable or {}
        self._bn_update = bn_update or {}
        self.bn_batch = self.training and bn_mode == "batch"
        self.dropout = bool(dropout) and self.training
        # Dropout mask = f(seed, step, element): every data-parallel rank draws its own masks, every step a new one
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.seed = (int(seed) + 0x632BE59BD9B4E019 * self.rank) & 0xFFFFFFFFFFFFFFFF
        if bn_variance not in BN_VARIANCE:
            raise ValueError("bn_variance must be one of %s" % sorted(BN_VARIANCE))
        self.bn_variance = bn_variance
        # data parallel: the loss normaliser count(w != 0) is the GLOBAL batch's (the reference's multi_gpu_model merges
        # the tower outputs and evaluates ONE loss, utils.py:209-211) — the host writes it before each step instead of
        # the in-graph dl3_count_nonzero of the local shard (Engine.set_nnz)
        self.external_nnz = bool(external_nnz)
        self.device = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.use_graph = use_graph
        self.fold_tail = True   # the loss kernel folds its gradient rows onto the low-resolution columns (the full-resolution dlogits never exist)
        self.dw_impl = dw_impl
        self.ops_prep, self.ops_fwd, self.ops_bwd = [], [], []
        # backward fork (round 3, opt-in: DL3_FORK=1): the 1x1-conv weight gradients are off the critical path until Adam —
        # they leave the backward chain as a second captured stream (event fork behind the launch that completes their dY,
        # one join at the end).  Measured SLOWER at every batch size (two 256-VGPR workgroups fill a CU), so
        # the default is one stream; kept because it is tested bit-identical and documents the experiment.
        self.poison = os.environ.get("DL3_POISON_SCRATCH", "0") == "1"
        # every weight-gradient slab fold of the backward pass in ONE launch at its end (DL3_BATCH_FOLDS=0: one
        # dl3_reduce_partials behind each weight-gradient launch, bit-identical — a test toggles it)
        self.batch_folds = os.environ.get("DL3_BATCH_FOLDS", "1") == "1"
        # ... of the folds whose partial rows are small: a deferred fold re-reads its slabs from HBM instead of the cache
        # they were just written through, which costs more than the launch it saves from ~12 MB on (Xception B=16 with
        # every fold deferred: 193.9 -> 196.4 ms)
        self.fold_defer_bytes = 8 << 20
        self.own_fold_ws_bytes = 0   # device memory of the deferred folds' own slab workspaces (reported by bench.py)
        self._folds = []
        # round 4: the weight-gradient launch of a BatchNorm'ed 1x1 convolution also writes dY = cA*g + cB*y + cC (it
        # assembles it anyway), and the bwd-data GEMM of the layer reads that ONE tensor (dl3_pwconv_bwd_weight_dy).  The
        # write is paid by the weight-gradient launch: cheap next to a wide X (project convolutions, +0.02-0.06 ms against
        # -0.3-0.7 ms of bwd-data at B=128) but not for an HBM-bound launch with a narrow X and a wide dY (16 -> 96 at
        # 256x256: +0.88 ms against -0.65): only where N <= K or K >= 96 (DL3_DY_MAT=0 disables)
        # (second look, same call: the criterion is the width of X against dY, not the width of dY: N <= K or K >= 96)
        self.dy_mat = os.environ.get("DL3_DY_MAT", "1") != "0"
        self.dy_mat_mink = 96
        # both gradients of an HBM-bound 1x1 convolution with a small weight matrix in one pass (dl3_pwconv_bwd_fused):
        # layers with at least DL3_FUSED_ROWS pixel rows (DL3_FUSED_BWD=0 disables)
        self.fused_bwd = os.environ.get("DL3_FUSED_BWD", "1") != "0"
        self.fused_min_rows = int(os.environ.get("DL3_FUSED_ROWS", "32768"))
        self.dy_buf = None
        self.fork = os.environ.get("DL3_FORK", "0") in ("1", "2")
        # DL3_FORK=2 (experiment): only the weight gradient that can run next to an HBM-bound depthwise backward launch
        # leaves the chain, and the chain's next GEMM waits for it: matrix-bound and HBM-bound kernels overlap, two GEMMs never do
        self.fork_pairs = os.environ.get("DL3_FORK", "0") == "2"
        self._join_before = set()   # id(op record): the main stream waits for the side stream before this launch
        self.bn_sites = []
        self.add_of_buf = {}        # id(Add output Buf) -> AddUnit
        self.prestat = {}           # id(Buf) -> (dpart, P, ld): BatchNorm-backward sums already reduced by a producer's epilogue
        self._side = set()          # id(op record) of the launches that run on the side stream
        self._side_stream = None
        self._fork_events = []
        self.side_scratch_bytes = 0
        self._side_scratch_users = []
        self.units = []
        self.views = {}
        self.bufs = []
        self.scratch_bytes = 0
        self.iteration = 0
        self.graph = None
        self._calls = 0
        self._keep = []
        self.drop_step = torch.zeros(1, dtype=torch.int64, device=self.device)  # device-side step number (dropout)
        self._plan_channels()
        self._build_params()
        self._lower()
        if self.training:
            self._lower_backward()
            if self.dropout:
                self.op(self.ops_bwd, "dl3_counter_add", self.drop_step.data_ptr(), 1)
        self.scratch = self.empty(max(self.scratch_bytes // 4, 4))
        for op in self._scratch_users:
            op[2][op[3]] = self.scratch.data_ptr()
        if self._side_scratch_users:  # the side stream's launches are ordered among themselves: one workspace of their own
            self.side_scratch = self.empty(max(self.side_scratch_bytes // 4, 4))
            for op in self._side_scratch_users:
                op[2][op[3]] = self.side_scratch.data_ptr()
        self.dirty = True

    # ------------------------------------------------------------------ memory
    def empty(self, n):
        t = torch.empty(int(n), dtype=torch.float32, device=self.device)
        if self.poison:   # DL3_POISON_SCRATCH=1 (test aid): a launch that reads scratch nobody wrote turns the loss into NaN
            t.fill_(float("nan"))
        self._keep.append(t)
        return t

    def zeros(self, n):
        t = torch.zeros(int(n), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def _plan_channels(self):
        self.phys = plan_channels(self.model)

    def _dshape(self, l, name, hshape):
        return device_shape(self.phys, l, name, hshape)

    @staticmethod
    def _pad_to(name, w, dshape):
        w = np.asarray(w, np.float32)
        if tuple(w.shape) == tuple(dshape):
            return w
        out = np.full(dshape, 1.0 if name.endswith("/moving_variance:0") else 0.0, np.float32)
        out[tuple(slice(0, n) for n in w.shape)] = w
        return out

    def _build_params(self):
        """Flat arenas: trainable parameters (kernels, biases, gamma, beta) | state (moving statistics and the
        weights of non-trainable layers).  One RCCL all-reduce and one Adam launch cover the whole model."""
        self.slots = {}
        self.hshape = {}
        np_, ns = 0, 0
        for l in self.model.layers:
            for name, w in l.weights.items():
                dshape = self._dshape(l, name, w.shape)
                size = int(np.prod(dshape))
                n = (size + 3) // 4 * 4
                self.hshape[name] = tuple(w.shape)
                if self._opt_trainable.get(l.name, l.trainable) and "/moving_" not in name:
                    self.slots[name] = ("p", np_, size, dshape, l)
                    np_ += n
                else:
                    self.slots[name] = ("s", ns, size, dshape, l)
                    ns += n
        self.n_param = np_
        self.params = self.zeros(max(np_, 4))
        self.state = self.zeros(max(ns, 4))
        # [count(w != 0) of the shard, loss sum of the shard, 0, 0] ride behind the gradients in the arena exchange
        # (data-parallel engines, Engine.train_step)
        self.tail = (max(np_, 4) + 3) // 4 * 4
        if self.training:
            self.grads = self.zeros(self.tail + 4)
            self.adam_m = self.zeros(max(np_, 4))
            self.adam_v = self.zeros(max(np_, 4))
            self.dummy_grad = self.zeros(4096)
        self.sync_all_to_device()

    def _arena(self, kind):
        return self.params if kind == "p" else self.state

    def wptr(self, name):
        kind, off, _, _, _ = self.slots[name]
        return self._arena(kind).data_ptr() + 4 * off

    def gptr(self, name, size=0):
        """gradient slot of a weight; non-trainable weights get a throw-away slot"""
        kind, off, n, _, _ = self.slots[name]
        if kind == "p":
            return self.grads.data_ptr() + 4 * off
        if n > self.dummy_grad.numel():
            self.dummy_grad = self.zeros(n)
        return self.dummy_grad.data_ptr()

    def trainable(self, name):
        return self.slots[name][0] == "p"

    def sync_all_to_device(self):
        hp = np.zeros(self.params.numel(), np.float32)
        hs = np.zeros(self.state.numel(), np.float32)
        for name, (kind, off, n, shp, l) in self.slots.items():
            (hp if kind == "p" else hs)[off:off + n] = self._pad_to(name, l.weights[name], shp).reshape(-1)
        self.params.copy_(torch.from_numpy(hp))
        self.state.copy_(torch.from_numpy(hs))
        self.dirty = True

    def sync_all_to_host(self):
        hp = self.params.cpu().numpy()
        hs = self.state.cpu().numpy()
        for name, (kind, off, n, shp, l) in self.slots.items():
            l.weights[name] = self._unpad(name, (hp if kind == "p" else hs)[off:off + n].reshape(shp))

    def sync_layer_to_host(self, layer):
        for name in layer.weights:
            kind, off, n, shp, _ = self.slots[name]
            layer.weights[name] = self._unpad(name, self._arena(kind)[off:off + n].cpu().numpy().reshape(shp))

    def sync_layer_to_device(self, layer):
        for name, w in layer.weights.items():
            kind, off, n, shp, _ = self.slots[name]
            self._arena(kind)[off:off + n].copy_(torch.from_numpy(np.ascontiguousarray(self._pad_to(name, w, shp).reshape(-1))))
        self.dirty = True

    def _unpad(self, name, arr):
        """device layout -> the Keras shape of the weight (drops the zero channels of _cp)"""
        return arr[tuple(slice(0, n) for n in self.hshape[name])].copy()

    def activate(self):
        """make this engine the one that layer.get_weights()/set_weights() talk to"""
        for l in self.model.layers:
            if l._engine is not None and l._engine is not self:
                pass
            l._engine = self

    # ------------------------------------------------------------------ op recording
    def op(self, lst, name, *args):
        fn = getattr(self.lib, name)
        rec = (name, fn, list(args), None)
        lst.append(rec)
        return rec

    def op_side(self, lst, name, *args):
        """op for a launch that may leave the backward chain (see self.fork) and brings its own workspace"""
        rec = self.op(lst, name, *args)
        if self.fork:
            self._side.add(id(rec))
        return rec

    def defer_fold(self, P, n):
        return self.batch_folds and 4 * int(P) * int(n) <= self.fold_defer_bytes

    def fold(self, src_ptr, P, n, dst_ptr):
        """dst[n] = sum over the P partial rows at src: now, or with every other fold in one launch (self.batch_folds)"""
        if self.defer_fold(P, n):
            self._folds.append((int(src_ptr), int(dst_ptr), int(P), int(n)))
        else:
            self.op(self.ops_bwd, "dl3_reduce_partials", src_ptr, P, n, dst_ptr)

    _scratch_users = None

    def op_ws(self, lst, name, nbytes, ws_index, *args):
        """op that needs the shared scratch workspace (pointer patched in once its size is known)"""
        self.scratch_bytes = max(self.scratch_bytes, int(nbytes))
        args = list(args)
        rec = (name, getattr(self.lib, name), args, ws_index)
        lst.append(rec)
        self._scratch_users.append(rec)
        return rec

    def op_ws_side(self, lst, name, nbytes, ws_index, *args):
        """op_ws for a launch that may leave the backward chain (see self.fork): its workspace is the side stream's"""
        if not self.fork:
            return self.op_ws(lst, name, nbytes, ws_index, *args)
        self.side_scratch_bytes = max(self.side_scratch_bytes, int(nbytes))
        rec = (name, getattr(self.lib, name), list(args), ws_index)
        lst.append(rec)
        self._side_scratch_users.append(rec)
        self._side.add(id(rec))
        return rec

    def run_ops_forked(self, lst):
        """lst with the launches marked in self._side on a second stream: each one waits for the event recorded on the
        main stream at its position in the plan (everything it reads is complete there and is never written again within
        the step), the main stream joins the side stream at the end.  Under hipGraph capture this becomes a parallel
        branch of the graph."""
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        mst, sst = main.cuda_stream, side.cuda_stream
        pending = True  # the main stream has moved on since the side stream last synchronised with it
        nev = 0
        for rec in lst:
            name, fn, args, _ = rec
            if id(rec) in self._join_before:
                main.wait_stream(side)
            if id(rec) in self._side:
                if pending:
                    if nev == len(self._fork_events):
                        self._fork_events.append(torch.cuda.Event())
                    ev = self._fork_events[nev]
                    nev += 1
                    ev.record(main)
                    side.wait_event(ev)
                    pending = False
                rc = fn(*args, sst)
            else:
                rc = fn(*args, mst)
                pending = True
            if rc != 0:
                capi.check(rc, name)
        main.wait_stream(side)

    def transpose(self, src_ptr, dst, rows, cols):
        """queue dst[cols][rows] = src[rows][cols]^T for the batched transpose at the head of the backward pass"""
        self._transposes.append((int(src_ptr), dst, int(rows), int(cols)))

    def run_ops(self, lst):
        st = torch.cuda.current_stream().cuda_stream
        for name, fn, args, _ in lst:
            rc = fn(*args, st)
            if rc != 0:
                capi.check(rc, name)

    # ------------------------------------------------------------------ forward lowering
    def _lower(self):
        self._scratch_users = []
        m = self.model
        topo = m._topo
        self.consumers = {}
        for l in topo:
            for t in (l.inbound if l is not m.input.layer else []):
                self.consumers.setdefault(id(t), []).append(l)
        self.placement = {}
        self.concat_bufs = {}
        self.bcast_resize = {}   # id(ResizeBilinear layer) -> the Concatenate it feeds as a per-image constant
        self.bcast_src = {}      # id(concat Buf) -> (View of the 1x1 source, channels)
        for l in topo:
            if l.kind == "Concatenate":
                self._place_concat(l)
        self.unit_of_buf = {}
        for l in topo:
            getattr(self, "_lo_" + l.kind)(l)
        out = self.views[id(m.output)]
        self.out_view = out
        self._check_centred_consumers()

    def _check_centred_consumers(self):
        """a slice stored as y - mean (centred frozen BatchNorm) must never be read raw: every unit input that touches it
        has to be an affine view (the view carries the matching scale / shift)"""
        for u in self.units:
            for w in (getattr(u, "inv", None), getattr(u, "a", None), getattr(u, "b", None)):
                if w is None or not w.buf.centred:
                    continue
                hit = any(off < w.off + w.C and w.off < off + C for _, off, C in w.buf.bns if off in w.buf.centred)
                if hit and not w.aff:
                    raise RuntimeError("unit %s reads the centred tensor %s raw" % (type(u).__name__, w.buf.name))

    def _producer_of(self, t):
        l = t.layer
        while l.kind in ("BatchNormalization", "Activation") and l.cfg.get("fn") != "softmax":
            l = l.inbound[0].layer
        return l

    def _bcast_input(self, l):
        """the first input of Concatenate l if it is the bilinear 'resize' of a 1x1 map — a per-image constant, e.g. the
        ASPP image-pooling branch (deeplabv3p.py:375-382) — and the Concatenate feeds exactly one 1x1 convolution: that
        branch is then never materialised; the convolution adds its contribution once per image (PwUnit img_add)"""
        t = l.inbound[0]
        p = self._producer_of(t)
        cs = self.consumers.get(id(l.output), [])
        if (p.kind == "ResizeBilinear" and p is t.layer and tuple(p.inbound[0].shape[:2]) == (1, 1) and len(l.inbound) > 1
                and len(cs) == 1 and cs[0].kind == "Conv2D" and cs[0].cfg["k"] == 1 and cs[0].cfg["stride"] == 1
                and len(self.consumers.get(id(t), [])) == 1):
            return p
        return None

    def _place_concat(self, l):
        H, W, C = l.output.shape
        bc = self._bcast_input(l)
        if bc is not None:
            self.bcast_resize[id(bc)] = l
            C -= l.inbound[0].shape[2]
        buf = Buf(self, self.B, H, W, C, l.name)
        self.bufs.append(buf)
        self.concat_bufs[id(l)] = buf
        off = 0
        for t in l.inbound:
            if bc is not None and t is l.inbound[0]:
                continue
            p = self._producer_of(t)
            ok = (p.kind in ("Conv2D",) and p.cfg["k"] == 1 and p.cfg["stride"] == 1) or p.kind == "ResizeBilinear"
            if p.kind == "Dropout" or not ok:
                raise NotImplementedError("Concatenate input produced by %s (%s) is not on the path" % (p.name, p.kind))
            self.placement[id(p)] = (buf, off)
            off += t.shape[2]

    def _new_out(self, l, H, W, C):
        """output buffer of a producing layer: a concat slice if placed, else a fresh Buf"""
        if id(l) in self.placement:
            buf, off = self.placement[id(l)]
            return buf, off
        buf = Buf(self, self.B, H, W, C, l.name)
        self.bufs.append(buf)
        return buf, 0

    def _in(self, l, i=0):
        return self.views[id(l.inbound[i])]

    def _consume(self, view):
        if view.buf.requires_grad:
            view.buf.expected += 1

    def _bn_follows(self, l):
        cs = self.consumers.get(id(l.output), [])
        return len(cs) == 1 and cs[0].kind == "BatchNormalization"

    def _want_stat(self, l, rows):
        """does the convolution's epilogue reduce BatchNorm partial sums?  (small tensors: dl3_bn_finalize_direct)"""
        return self.bn_batch and self._bn_follows(l) and rows > SMALL_BN_ROWS

    def _stat_buf(self, P, C):
        return self.empty(P * C * 2)

    # ---- layers ---------------------------------------------------------------------
    def _lo_InputLayer(self, l):
        H, W, C = l.output.shape
        self.xbuf = Buf(self, self.B, H, W, C, "input", requires_grad=False)
        self.bufs.append(self.xbuf)
        self.views[id(l.output)] = View(self.xbuf, 0, C)

    def _lo_Prescale(self, l):
        v = self._in(l)
        assert v.buf is self.xbuf and not v.aff
        vec = v.buf.vec.cpu()
        vec[V_SCALE] = 1.0 / 127.5
        vec[V_SHIFT] = -1.0
        v.buf.vec.copy_(vec)
        self.views[id(l.output)] = v.derive(aff=True)

    def _lo_ZeroPadding2D(self, l):
        v = self._in(l)
        (pt, pb), (pl, pr) = l.cfg["pad"]
        assert v.pad is None
        self.views[id(l.output)] = v.derive(pad=(pt, pl))

    def _lo_Activation(self, l):
        v = self._in(l)
        fn = l.cfg["fn"]
        if fn == "softmax":
            return self._lo_softmax(l, v)
        if v.act != ACT_NONE:
            # relu(relu(x)) == relu(x), relu(relu6(x)) == relu6(x), relu6(relu(x)) == relu6(x) — values and gradient
            # masks alike (Xception: entry_flow_block1 re-applies ReLU to an already rectified tensor)
            act = ACT_RELU6 if ACT_RELU6 in (v.act, _ACT[fn]) else ACT_RELU
            self.views[id(l.output)] = v.derive(act=act)
            return
        self.views[id(l.output)] = v.derive(act=_ACT[fn])

    def _lo_Reshape(self, l):
        self.views[id(l.output)] = self._in(l)

    def _lo_BatchNormalization(self, l):
        v = self._in(l)
        unit = self.unit_of_buf.get((id(v.buf), v.off))
        if v.aff or v.act != ACT_NONE or unit is None or unit.bn is not None:
            raise NotImplementedError("BatchNormalization %s must directly follow a convolution" % l.name)
        unit.bn = l
        C = v.C
        n = l.name
        v.buf.bns.append((l, v.off, C))
        if self.bn_batch:
            # Keras 2.2.x collects no updates from a non-trainable layer: a frozen BatchNormalization still normalises
            # with the batch statistics in the training phase, but its moving statistics stay as loaded
            upd = self._bn_update.get(l.name, l.trainable)
            if v.buf.M <= SMALL_BN_ROWS:
                # few rows (the image-pooling branch: one row per image): two-pass statistics straight from the tensor
                self.op(self.ops_fwd, "dl3_bn_finalize_direct", v.p(), v.ld, v.buf.M, C,
                        self.wptr(n + "/gamma:0"), self.wptr(n + "/beta:0"), fused_bn_epsilon(l.cfg["eps"]), l.cfg["momentum"],
                        BN_VARIANCE[self.bn_variance](float(v.buf.M), float(l.cfg["eps"])),
                        v.buf.vptr(V_SCALE, v.off), v.buf.vptr(V_SHIFT, v.off), v.buf.vptr(V_MEAN, v.off),
                        v.buf.vptr(V_INVSTD, v.off), self.wptr(n + "/moving_mean:0") if upd else None,
                        self.wptr(n + "/moving_variance:0") if upd else None)
                self.views[id(l.output)] = v.derive(aff=True)
                return
            self.bn_sites.append((len(self.ops_fwd), l, v.buf, v.off, C, unit))  # (index in ops_fwd, ...): diagnostics
            self.op(self.ops_fwd, "dl3_bn_finalize", ptr(unit.stat), unit.P, C, C, float(v.buf.M),
                    self.wptr(n + "/gamma:0"), self.wptr(n + "/beta:0"), fused_bn_epsilon(l.cfg["eps"]), l.cfg["momentum"],
                    BN_VARIANCE[self.bn_variance](float(v.buf.M), float(l.cfg["eps"])),
                    v.buf.vptr(V_SCALE, v.off), v.buf.vptr(V_SHIFT, v.off), v.buf.vptr(V_MEAN, v.off),
                    v.buf.vptr(V_INVSTD, v.off), self.wptr(n + "/moving_mean:0") if upd else None,
                    self.wptr(n + "/moving_variance:0") if upd else None)
        elif isinstance(unit, PwUnit) and unit.bias is None:
            # moving statistics, 1x1 convolution in front (round 4): the GEMM's free bias slot takes -mean, the tensor holds
            # y - mean and consumers read scale*(y - mean) + beta — the reference's own gamma*(x - mean)/sqrt(var + eps) +
            # beta instead of scale*y + (beta - mean*scale), whose shift carries half an ulp of |mean*scale| into every
            # element (dl3_bn_frozen_centered)
            negm = v.buf.vptr(V_NEGMEAN, v.off)
            self.op(self.ops_prep, "dl3_bn_frozen_centered", self.wptr(n + "/gamma:0"), self.wptr(n + "/beta:0"),
                    self.wptr(n + "/moving_mean:0"), self.wptr(n + "/moving_variance:0"), fused_bn_epsilon(l.cfg["eps"]), C,
                    v.buf.vptr(V_SCALE, v.off), v.buf.vptr(V_SHIFT, v.off), v.buf.vptr(V_MEAN, v.off),
                    v.buf.vptr(V_INVSTD, v.off), negm)
            unit.set_output_offset(negm)   # (the buffer holds y - mean from here on: Buf.centred)
        else:
            self.op(self.ops_prep, "dl3_bn_frozen", self.wptr(n + "/gamma:0"), self.wptr(n + "/beta:0"),
                    self.wptr(n + "/moving_mean:0"), self.wptr(n + "/moving_variance:0"), fused_bn_epsilon(l.cfg["eps"]), C,
                    v.buf.vptr(V_SCALE, v.off), v.buf.vptr(V_SHIFT, v.off), v.buf.vptr(V_MEAN, v.off),
                    v.buf.vptr(V_INVSTD, v.off))
        self.views[id(l.output)] = v.derive(aff=True)

    def _lo_Conv2D(self, l):
        v = self._in(l)
        if l.cfg["k"] == 3:
            return self._lo_conv3x3(l, v)
        if l.cfg["stride"] != 1:
            # Xception shortcut (deeplabv3p.py:143-145): compact the sampled pixels, then the ordinary GEMM
            st = l.cfg["stride"]
            Ho, Wo = l.output.shape[0], l.output.shape[1]
            cbuf = Buf(self, self.B, Ho, Wo, v.C, l.name + "_sub")
            self.bufs.append(cbuf)
            self.units.append(SubsampleUnit(self, v, View(cbuf, 0, v.C), st))
            v = View(cbuf, 0, v.C)
        Ho, Wo, N = v.shape[1], v.shape[2], self.phys[id(l.output)]
        buf, off = self._new_out(l, Ho, Wo, N)
        assert id(l) in self.placement or buf.ld == N, (l.name, buf.ld, N)  # an own buffer is exactly as wide as stored
        if id(v.buf) in self.bcast_src:
            # Concatenate([broadcast of a 1x1 map, per-pixel branches]) -> 1x1 convolution: rows 0..Cb-1 of the kernel meet a
            # per-image constant: a GEMM of B rows whose result is added once per image; the other rows run per pixel
            sv, Cb = self.bcast_src[id(v.buf)]
            if v.off != 0 or v.C != v.buf.ld:
                raise NotImplementedError("convolution over a slice of a Concatenate with a broadcast input")
            ib = Buf(self, self.B, 1, 1, N, l.name + "_per_image")
            self.bufs.append(ib)
            pu = PwUnit(self, l, sv, View(ib, 0, N), want_stat=False, wrow0=0, use_bias=False)
            self.units.append(pu)
            u = PwUnit(self, l, v, View(buf, off, N), want_stat=self._want_stat(l, buf.M), wrow0=Cb, img_add=ib)
            u.img_unit = pu
        else:
            u = PwUnit(self, l, v, View(buf, off, N), want_stat=self._want_stat(l, buf.M))
        self._register(u, buf, off)
        self.views[id(l.output)] = View(buf, off, N)

    def _lo_Subpixel(self, l):
        v = self._in(l)
        r, co = l.cfg["r"], l.cfg["out_filters"]
        k = l.cfg["k"]
        if l.cfg["stride"] != 1:
            raise NotImplementedError("Subpixel with strides != 1 (%s)" % l.name)
        if k != 1:
            # subpixel.py:42-58: any kernel_size.  The k x k taps of the input are gathered side by side (dl3_conv_taps_fwd)
            # and the Keras kernel [k][k][Cin][F], read as a [k*k*Cin][F] matrix, drives the same GEMM as the 1x1 case
            v = self._taps(l, v, k)
        Ho, Wo, N = v.shape[1], v.shape[2], l.cfg["filters"]
        buf = Buf(self, self.B, Ho, Wo, N, l.name)
        self.bufs.append(buf)
        u = PwUnit(self, l, v, View(buf, 0, N), want_stat=False)
        self._register(u, buf, 0)
        obuf = Buf(self, self.B, Ho * r, Wo * r, co, l.name + "_shift")
        self.bufs.append(obuf)
        s = ShuffleUnit(self, View(buf, 0, N), View(obuf, 0, co), r, co)
        self.units.append(s)
        self.views[id(l.output)] = View(obuf, 0, co)

    def _taps(self, l, v, k):
        N_, H, W, C = v.shape
        if l.cfg["padding"] == "same":
            Ho, pt = _same_pads(H, k, 1, 1)
            Wo, pl = _same_pads(W, k, 1, 1)
        else:
            Ho, Wo, pt, pl = H - k + 1, W - k + 1, 0, 0
        cbuf = Buf(self, self.B, Ho, Wo, k * k * C, l.name + "_taps")
        self.bufs.append(cbuf)
        out = View(cbuf, 0, k * k * C)
        self.units.append(TapsUnit(self, v, out, k, pt, pl))
        return out

    def _lo_conv3x3(self, l, v):
        N_, H, W, Cin = v.shape
        s = l.cfg["stride"]
        if l.cfg["padding"] == "same":
            Ho, pt = _same_pads(H, 3, s, 1)
            Wo, pl = _same_pads(W, 3, s, 1)
        else:
            pt, pl = v.pad or (0, 0)
            Ho, Wo = l.output.shape[0], l.output.shape[1]
        assert self.phys[id(l.output)] == l.cfg["filters"], "dense 3x3 outputs are stored at their logical width"
        buf, off = self._new_out(l, Ho, Wo, l.cfg["filters"])
        u = Conv3Unit(self, l, v, View(buf, off, l.cfg["filters"]), s, pt, pl,
                      want_stat=self._want_stat(l, buf.M))
        self._register(u, buf, off)
        self.views[id(l.output)] = View(buf, off, l.cfg["filters"])

    def _lo_DepthwiseConv2D(self, l):
        v = self._in(l)
        N_, H, W, C = v.shape
        if v.off != 0 or v.ld != C:
            raise NotImplementedError("depthwise conv on a channel slice (%s)" % l.name)
        s, r = l.cfg["stride"], l.cfg["rate"]
        if l.cfg["padding"] == "same":
            Ho, pt = _same_pads(H, 3, s, r)
            Wo, pl = _same_pads(W, 3, s, r)
        else:
            pt, pl = v.pad or (0, 0)
            Ho, Wo = l.output.shape[0], l.output.shape[1]
        buf, off = self._new_out(l, Ho, Wo, C)
        u = DwUnit(self, l, v, View(buf, off, C), s, r, pt, pl, want_stat=self._want_stat(l, buf.M))
        self._register(u, buf, off)
        self.views[id(l.output)] = View(buf, off, C)

    def _register(self, u, buf, off):
        self.units.append(u)
        self.unit_of_buf[(id(buf), off)] = u

    def _lo_Add(self, l):
        a, b = self._in(l, 0), self._in(l, 1)
        H, W, C = l.output.shape
        C = self.phys[id(l.output)]
        buf = Buf(self, self.B, H, W, C, l.name)
        self.bufs.append(buf)
        self.units.append(AddUnit(self, a, b, View(buf, 0, C)))
        self.add_of_buf[id(buf)] = self.units[-1]
        self.views[id(l.output)] = View(buf, 0, C)

    def _lo_Concatenate(self, l):
        buf = self.concat_bufs[id(l)]
        acts = set()
        for t in l.inbound:
            v = self.views[id(t)]
            if v is None and id(buf) in self.bcast_src:
                sv = self.bcast_src[id(buf)][0]
                if sv.act != ACT_NONE:
                    acts.add(sv.act)
                continue
            if v.buf is not buf:
                raise NotImplementedError("concat input %s was not placed" % t.layer.name)
            if v.act != ACT_NONE:
                acts.add(v.act)
            elif not (v.buf.nonneg_slices.get(v.off, False) if hasattr(v.buf, "nonneg_slices") else False):
                raise NotImplementedError("concat input without activation must be known non-negative")
        if len(acts) > 1:
            raise NotImplementedError("concat inputs with different activations")
        act = acts.pop() if acts else ACT_NONE
        self.views[id(l.output)] = View(buf, 0, buf.ld, act=act, aff=True)

    def _lo_AveragePooling2D(self, l):
        v = self._in(l)
        C = v.C
        buf = Buf(self, self.B, 1, 1, C, l.name)
        self.bufs.append(buf)
        self.units.append(GapUnit(self, v, View(buf, 0, C)))
        self.views[id(l.output)] = View(buf, 0, C)

    def _lo_ResizeBilinear(self, l):
        v = self._in(l)
        if id(l) in self.bcast_resize:
            cat = self.bcast_resize[id(l)]
            self.bcast_src[id(self.concat_bufs[id(cat)])] = (v, v.C)
            self.views[id(l.output)] = None  # never materialised (Engine._bcast_input)
            return
        Ho, Wo = l.cfg["size"]
        buf, off = self._new_out(l, Ho, Wo, v.C)
        nonneg = v.act in (ACT_RELU, ACT_RELU6) or v.buf.nonneg
        if buf.ld != v.C:
            if not hasattr(buf, "nonneg_slices"):
                buf.nonneg_slices = {}
            buf.nonneg_slices[off] = nonneg
        else:
            buf.nonneg = nonneg
        self.units.append(ResizeUnit(self, v, View(buf, off, v.C), Ho, Wo))
        self.views[id(l.output)] = View(buf, off, v.C)

    def _lo_Dropout(self, l):
        v = self._in(l)
        if not (self.dropout and l.cfg["rate"] > 0):
            self.views[id(l.output)] = v
            return
        N_, H, W, C = v.shape
        buf = Buf(self, self.B, H, W, C, l.name)
        buf.nonneg = v.act in (ACT_RELU, ACT_RELU6)
        self.bufs.append(buf)
        self.units.append(MaterializeUnit(self, v, View(buf, 0, C), l.cfg["rate"], self.seed))
        self.views[id(l.output)] = View(buf, 0, C)

    def _lo_softmax(self, l, v):
        N_, H, W, C = v.shape[0], v.buf.H, v.buf.W, v.C
        if v.aff or v.act != ACT_NONE or v.off != 0 or v.ld != C:
            raise NotImplementedError("softmax expects materialised logits")
        self.logits_view = v
        # training: fuse the final bilinear resize into the loss kernel (the full-resolution logits are interpolated in
        # registers, never written): drop the resize launch from the forward plan, keep it for logits() on demand
        self.fused_tail = None
        last = self.units[-1] if self.units else None
        if (self.training and isinstance(last, ResizeUnit) and last.outv.buf is v.buf and C <= 32
                and not last.inv.aff and last.inv.act == ACT_NONE and last.inv.off == 0 and last.inv.ld == C):
            self.fused_tail = last
            self._resize_op = self.ops_fwd.pop()
            assert self._resize_op[0] == "dl3_resize_bilinear_fwd"
        # Subpixel head in training: the loss is evaluated straight on the unshuffled output of the Subpixel convolution
        # (dl3_shuffle_softmax_xent): neither phase-shift pass nor the shuffled logits / gradient touch HBM; the forward
        # phase shift is kept for logits() on demand
        self.fused_shuffle = None
        if (self.training and self.fused_tail is None and isinstance(last, ShuffleUnit) and last.outv.buf is v.buf
                and os.environ.get("DL3_FUSE_SHUFFLE", "1") != "0"
                # the fused kernel reads the Subpixel convolution's output as a plain contiguous tensor of co*r*r channels
                # and writes that buffer's gradient itself: only with a single consumer and no view on top
                and not last.inv.aff and last.inv.act == ACT_NONE and last.inv.off == 0
                and last.inv.ld == last.co * last.r * last.r and last.inv.buf.expected == 1 and not last.inv.buf.bns
                and self.lib.dl3_shuffle_xent_partials(self.B, last.inv.buf.H, last.inv.buf.W, last.co, last.r) > 0 and C <= 32):
            self.fused_shuffle = last
            self._shuffle_op = self.ops_fwd.pop()
            assert self._shuffle_op[0] == "dl3_phase_shift"
        M = v.buf.M
        self.probs = self.empty(M * C)
        self.out_shape = (self.B,) + tuple(l.output.shape)
        if self.training:
            self.labels = self.zeros(M)
            self.sweights = self.zeros(M)
            self.nnz = self.zeros(4)
            self.lossP = self.lib.dl3_rows_partials(M)
            self.loss_part = self.zeros(self.lossP)
            if self.external_nnz:
                # data parallel: the loss kernel divides by a FIXED normaliser c0 (pixels per image: the same on every
                # rank, whatever the shard), the shard's count(w != 0) and its loss sum land in the arena's tail, the ONE
                # all-reduce of the step sums them with the gradients, and Adam finishes the scale c0 / count_all on the
                # device (dl3_adam_step_norm): no all-reduce in front of the step, no host round trip behind it
                self.nnz_host = float(M // self.B)
                self.nnz.fill_(self.nnz_host)
                self.loss = self.grads[self.tail + 1:self.tail + 2]
            else:
                self.loss = self.zeros(4)
            v.buf.expected += 1
        self.views[id(l.output)] = v

    # ------------------------------------------------------------------ backward lowering
    def _prune_gradients(self):
        """which buffers need a gradient at all: those computed from at least one TRAINABLE parameter.  With every layer
        trainable that is everything but the image; with the notebook's fine-tuning (json 147-155: `l.trainable = False`
        up to `concat_projection`) it is the tail only, and the backward pass below it — the whole backbone and the ASPP
        branches — is never lowered: no data gradient would reach a trainable weight through it (round 4; before, the
        frozen layers merely skipped their weight gradients).  DL3_PRUNE_BWD=0 restores that."""
        if os.environ.get("DL3_PRUNE_BWD", "1") == "0":
            return
        for b in self.bufs:
            b.requires_grad = False
        for u in self.units:   # forward order: producers before consumers
            ins = [w.buf for w in (getattr(u, "inv", None), getattr(u, "a", None), getattr(u, "b", None)) if w is not None]
            if getattr(u, "img_add", None) is not None:
                ins.append(u.img_add)
            own = False
            if isinstance(u, _ConvBase):
                names = [u.wname()] + ([u.bias] if getattr(u, "bias", None) else [])
                if u.bn is not None:
                    names += [u.bn.name + "/gamma:0", u.bn.name + "/beta:0"]
                own = any(self.trainable(n) for n in names)
            if own or any(b.requires_grad for b in ins):
                u.outv.buf.requires_grad = True   # (a Concatenate buffer: any of its producers)

    def _lower_backward(self):
        self._prune_gradients()
        v = self.logits_view
        buf = v.buf
        M, C = buf.M, v.C
        # loss + dlogits (first and only contribution to the logits buffer)
        self.op(self.ops_fwd, "dl3_count_nonzero", ptr(self.sweights), M,
                (self.grads.data_ptr() + 4 * self.tail) if self.external_nnz else ptr(self.nnz))
        if self.fused_shuffle is not None:
            su = self.fused_shuffle
            ub = su.inv.buf
            ub.grad = self.empty(ub.M * ub.ld)
            self.lossP = self.lib.dl3_shuffle_xent_partials(self.B, ub.H, ub.W, su.co, su.r)
            self.loss_part = self.zeros(self.lossP)
            self.op(self.ops_fwd, "dl3_shuffle_softmax_xent", su.inv.p(), ptr(self.labels), ptr(self.sweights), ptr(self.nnz),
                    ptr(ub.grad), ptr(self.loss_part), self.B, ub.H, ub.W, su.co, su.r)
            su.fused = True
        elif self.fused_tail is not None and ((buf.W + 2 * self.fused_tail.inv.buf.W) * C + 2 * buf.W) * 4 <= 65536 and self.fold_tail:
            # the full-resolution gradient never exists: the loss kernel folds each output row onto the low-resolution
            # columns, the resize unit's backward folds the rows
            lo = self.fused_tail.inv
            self.lossP = self.lib.dl3_xent_fold_partials(self.B, buf.H)
            self.loss_part = self.zeros(self.lossP)
            self.fused_tail.xfold = self.empty(self.B * buf.H * lo.buf.W * C)
            self.op(self.ops_fwd, "dl3_upsample_softmax_xent_fold", lo.p(), ptr(self.labels), ptr(self.sweights),
                    ptr(self.nnz), ptr(self.fused_tail.xfold), ptr(self.loss_part), self.B, lo.buf.H, lo.buf.W, buf.H,
                    buf.W, C)
        elif self.fused_tail is not None:
            buf.grad = self.empty(M * buf.ld)
            lo = self.fused_tail.inv
            self.op(self.ops_fwd, "dl3_upsample_softmax_xent", lo.p(), ptr(self.labels), ptr(self.sweights),
                    ptr(self.nnz), None, ptr(buf.grad), ptr(self.loss_part), self.B, lo.buf.H, lo.buf.W, buf.H, buf.W, C)
        else:
            buf.grad = self.empty(M * buf.ld)
            self.op(self.ops_fwd, "dl3_softmax_xent", ptr(buf.t), ptr(self.labels), ptr(self.sweights), ptr(self.nnz),
                    None, ptr(buf.grad), ptr(self.loss_part), M, C)
        self.op(self.ops_fwd, "dl3_reduce_partials", ptr(self.loss_part), self.lossP, 1, ptr(self.loss))
        buf.done = 1
        assert buf.expected == 1
        # weight transposes for bwd-data, then the units in reverse
        self._transposes = []
        first_bwd = len(self.ops_bwd)
        # one buffer holds the materialised dY of whichever 1x1 convolution is being differentiated (written by its
        # weight-gradient launch, read by its bwd-data launch right behind it on the same stream)
        n = max([u.M * u.N for u in self.units if isinstance(u, PwUnit) and u.dy_mat_ok()] or [0])
        if n:
            self.dy_buf = self.empty(n)
        for u in reversed(self.units):
            if u.outv.buf.requires_grad:   # (else: no trainable parameter at or above this unit)
                u.bwd()
        if self._transposes:
            # every W -> WT of the backward pass in ONE launch at its start (54 tiny launches otherwise)
            rows, t0 = [], 0
            for src, dst, r, c in self._transposes:
                tx = (c + 31) // 32
                rows.append([src, dst.data_ptr(), r, c, t0, tx])
                t0 += tx * ((r + 31) // 32)
            self._tdesc = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self._keep.append(self._tdesc)
            rec = ("dl3_transpose_batched", getattr(self.lib, "dl3_transpose_batched"),
                   [self._tdesc.data_ptr(), len(rows), t0], None)
            self.ops_bwd.insert(first_bwd, rec)
        if self._folds:
            # (the slabs were written by launches on either stream: the fold sits behind the join at the end of the list)
            rows, b0 = [], 0
            for src, dst, P, n in self._folds:
                rows.append([src, dst, P, n, b0])
                b0 += self.lib.dl3_reduce_partials_blocks(P, n)
            self._fdesc = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self._keep.append(self._fdesc)
            rec = self.op(self.ops_bwd, "dl3_reduce_partials_batched", self._fdesc.data_ptr(), len(rows), b0)
            if self.fork:
                self._join_before.add(id(rec))
        if self.fork_pairs:
            self._pair_wgrads_with_depthwise()
        self._check_dy_adjacency()
        assert not self.prestat, "BatchNorm-backward sums reduced early but never folded"
        for b in self.bufs:
            if b.requires_grad and b.expected and b.done != b.expected:
                raise RuntimeError("gradient accounting broken for buffer %s (%d/%d)" % (b.name, b.done, b.expected))

    def _check_dy_adjacency(self):
        """ONE buffer holds the materialised dY of whichever 1x1 convolution is being differentiated: between the launch
        that writes it (dl3_pwconv_bwd_weight_dy) and the layer's bwd-data launch that reads it, no other launch may write
        it — i.e. no second dl3_pwconv_bwd_weight_dy — and both must sit on the same stream (any future
        reordering of ops_bwd would otherwise corrupt gradients silently)"""
        if self.dy_buf is None:
            return
        dy = ptr(self.dy_buf)
        pending = None
        for rec in self.ops_bwd:
            name, args = rec[0], rec[2]
            if name == "dl3_pwconv_bwd_weight_dy":
                assert pending is None, "two dY writers without the first one's reader in between"
                assert args[-2] == dy and id(rec) not in self._side, "dY written off the main stream / into another buffer"
                pending = (args[14], args[15], args[16])    # M, K, N
            elif name == "dl3_pwconv_bwd_data" and args[0] == dy:
                assert pending == (args[22], args[23], args[24]), ("bwd-data reads a dY another layer wrote", pending)
                pending = None
        assert pending is None, "a materialised dY was never read"

    def _pair_wgrads_with_depthwise(self):
        """DL3_FORK=2: every depthwise backward launch takes the nearest earlier 1x1 weight gradient with it (moved to just
        in front of it, on the side stream); every other weight gradient stays in the chain; the next 1x1 GEMM of the chain
        waits for the side stream."""
        ops = self.ops_bwd
        side_all = [r for r in ops if id(r) in self._side]
        paired = set()
        j = 0
        while j < len(ops):
            if ops[j][0] == "dl3_dwconv3x3_bwd":
                i = j - 1
                while i >= 0 and not (ops[i][0] == "dl3_pwconv_bwd_weight" and id(ops[i]) in self._side and id(ops[i]) not in paired):
                    if ops[i][0] == "dl3_dwconv3x3_bwd":
                        i = -1
                        break
                    i -= 1
                if i >= 0:
                    rec = ops.pop(i)
                    ops.insert(j - 1, rec)   # directly in front of the depthwise launch (behind it measured no gain at all)
                    paired.add(id(rec))
            j += 1
        for p, rec in enumerate(ops):
            if id(rec) in paired:
                k = p + 1
                while k < len(ops) and (id(ops[k]) in paired or not ops[k][0].startswith("dl3_pwconv_")):
                    k += 1
                if k < len(ops):
                    self._join_before.add(id(ops[k]))
        for r in side_all:
            if id(r) not in paired:
                self._side.discard(id(r))

    # ---- gradient contribution protocol ---------------------------------------------
    def contrib_kernel(self, buf):
        """A kernel is about to write its contribution into buf.grad.  Returns (gout, add, last)."""
        add = None
        if buf.grad is None:
            buf.grad = self.empty(buf.M * buf.ld)
            if buf.addend is not None:
                add, buf.addend = buf.addend, None
        else:
            add = buf.grad
        buf.done += 1
        return buf.grad, add, buf.done == buf.expected

    def finish_bn_bwd(self, buf, dpart, P, ldc):
        """after the last contribution: fold the BN-backward partials of every BN living in this buffer"""
        for bn, off, C in buf.bns:
            n = bn.name
            self.op(self.ops_bwd, "dl3_bn_bwd_finalize", dpart.data_ptr() + 4 * 2 * off, P, ldc, C, float(buf.M),
                    self.wptr(n + "/gamma:0"), buf.vptr(V_MEAN, off), buf.vptr(V_INVSTD, off),
                    1 if self.bn_batch else 0, buf.vptr(V_CA, off), buf.vptr(V_CB, off), buf.vptr(V_CC, off),
                    self.gptr(n + "/gamma:0"), self.gptr(n + "/beta:0"))

    def contrib_elementwise(self, buf, view, gin, ldgin, gin_div=1, gin_scale=1.0, drop=(0.0, 0)):
        """contribution through dl3_grad_finish: gout = mask_view(gin) + existing; handles stats when last"""
        gout, add, last = self.contrib_kernel(buf)
        need_stat = last and bool(buf.bns)
        P = self.lib.dl3_rows_partials(buf.M)
        dpart = self.empty(P * buf.ld * 2) if need_stat else None
        masked = view.act != ACT_NONE
        need_x = masked or need_stat
        if view.off != 0 or view.C != buf.ld:
            raise NotImplementedError("element-wise gradient into a channel slice")
        self.op(self.ops_bwd, "dl3_grad_finish", gin, ldgin, gin_div, gin_scale, ptr(gout), buf.ld,
                ptr(add), buf.ld, ptr(buf.t) if need_x else None, buf.ld,
                view.scale() if masked else None, view.shift() if masked else None, view.act,
                buf.vptr(V_MEAN) if need_stat else None, buf.vptr(V_INVSTD) if need_stat else None, ptr(dpart),
                buf.M, buf.ld, drop[0], drop[1], self.drop_step.data_ptr() if drop[0] > 0 else None)
        if need_stat:
            self.finish_bn_bwd(buf, dpart, P, buf.ld)

    def contrib_passthrough(self, buf, view, g):
        """Add backward: the gradient tensor g (same [M, C] shape, ld == C) flows unchanged into buf"""
        if not buf.requires_grad:
            return
        if view.act != ACT_NONE or view.off != 0 or view.C != buf.ld or g.numel() != buf.M * buf.ld:
            return self.contrib_elementwise(buf, view, ptr(g), view.C)
        if buf.grad is None and buf.addend is None and buf.done == 0:
            buf.done += 1
            if buf.expected == 1:
                buf.grad = g  # alias
                if buf.bns and id(buf) in self.prestat:
                    # the kernel that completed g already reduced sum(g), sum(g * x_hat) of THIS BatchNorm in its epilogue
                    dpart, P, ld = self.prestat.pop(id(buf))
                    self.finish_bn_bwd(buf, dpart, P, ld)
                elif buf.bns:
                    P = self.lib.dl3_rows_partials(buf.M)
                    dpart = self.empty(P * buf.ld * 2)
                    self.op(self.ops_bwd, "dl3_grad_finish", ptr(g), buf.ld, 1, 1.0, ptr(g), buf.ld, None, 0,
                            ptr(buf.t), buf.ld, None, None, ACT_NONE, buf.vptr(V_MEAN), buf.vptr(V_INVSTD),
                            ptr(dpart), buf.M, buf.ld, 0.0, 0, None)
                    self.finish_bn_bwd(buf, dpart, P, buf.ld)
            else:
                buf.addend = g
            return
        self.contrib_elementwise(buf, view, ptr(g), buf.ld)

    def alias_stats_target(self, ibuf):
        """ibuf is the output of a residual Add whose gradient is about to be completed by a GEMM epilogue.  The Add's
        backward hands that very tensor on, unchanged, to its inputs; for an input that is a BatchNorm'ed convolution
        output with no other consumer (the project conv of the block, deeplabv3p.py:194-201) the BatchNorm-backward sums
        over the gradient can be reduced right there instead of by a separate dl3_grad_finish pass over g and y
        (round 3).  Returns that input's Buf or None."""
        add = self.add_of_buf.get(id(ibuf))
        if add is None:
            return None
        hits = [v for v in (add.a, add.b)
                if v.buf.requires_grad and v.buf.bns and v.buf.expected == 1 and v.buf.done == 0 and v.buf.grad is None
                and v.buf.addend is None and v.act == ACT_NONE and v.off == 0 and v.C == v.buf.ld == ibuf.ld
                and v.buf.M == ibuf.M and len(v.buf.bns) == 1 and v.buf.bns[0][1] == 0 and v.buf.bns[0][2] == v.buf.ld]
        return hits[0].buf if len(hits) == 1 else None

    def grad_operand(self, out_view):
        """(g, ldg, yraw, ldy, cA, cB, cC) of a conv whose output is out_view"""
        buf = out_view.buf
        g = buf.grad.data_ptr() + 4 * out_view.off
        has_bn = any(off == out_view.off for _, off, _ in buf.bns)
        if has_bn:
            return (g, buf.ld, out_view.p(), buf.ld, buf.vptr(V_CA, out_view.off), buf.vptr(V_CB, out_view.off),
                    buf.vptr(V_CC, out_view.off))
        return (g, buf.ld, None, 0, None, None, None)

    # ------------------------------------------------------------------ running
    def _prep(self):
        if self.dirty:
            self.run_ops(self.ops_prep)
            self.dirty = False

    def set_input(self, x):
        """raw 0-255 pixels [B,H,W,3]: float32 (the reference's arrays), or uint8 (decoded images as they come from
        cv2.imread) — those cross PCIe as bytes and are widened on the device"""
        if torch.is_tensor(x):
            xt = x
        elif isinstance(x, np.ndarray) and x.dtype == np.uint8:
            xt = torch.from_numpy(np.ascontiguousarray(x))
        else:
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        self.xbuf.t.copy_(xt.reshape(-1).to(self.device, non_blocking=True))

    def forward(self):
        """forward launch sequence; the inference engine replays it as one hipGraph from the second call on (68
        launches of a few microseconds each: at batch 1 the Python/ctypes launch path costs more than the kernels)"""
        self._prep()
        if self.training or not self.use_graph:
            self.run_ops(self.ops_fwd)
            return
        if self._calls >= 1:
            if self.graph is None:
                torch.cuda.synchronize()
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.run_ops(self.ops_fwd)
                    self.graph = g
                except Exception as e:  # pragma: no cover - depends on the runtime
                    print("dl3: hipGraph capture failed (%s); running eagerly" % e)
                    self.use_graph = False
                    torch.cuda.synchronize()
                    self.run_ops(self.ops_fwd)
                    return
            self.graph.replay()
        else:
            self.run_ops(self.ops_fwd)
        self._calls += 1

    def predict(self, x):
        assert not self.training
        self.set_input(x)
        self.forward()
        v = self.logits_view
        capi.call("dl3_softmax_fwd", ptr(v.buf.t), ptr(self.probs), v.buf.M, v.C,
                  torch.cuda.current_stream().cuda_stream)
        return self.probs.cpu().numpy().reshape(self.out_shape)

    def logits(self):
        v = self.logits_view
        if getattr(self, "fused_tail", None) is not None:
            self.run_ops([self._resize_op])  # the fused training tail never materialises them
        if getattr(self, "fused_shuffle", None) is not None:
            self.run_ops([self._shuffle_op])  # nor does the fused Subpixel tail
        return v.buf.t.cpu().numpy().reshape(self.B, v.buf.H, v.buf.W, v.C)

    def argmax(self):
        v = self.logits_view
        out = torch.empty(v.buf.M, dtype=torch.int32, device=self.device)
        capi.call("dl3_argmax", ptr(v.buf.t), out.data_ptr(), v.buf.M, v.C, torch.cuda.current_stream().cuda_stream)
        return out.cpu().numpy().reshape(self.B, v.buf.H, v.buf.W)

    def set_targets(self, y, sw=None):
        """labels [B,HW,1] (void = classes) and temporal sample weights [B,HW]; numpy arrays or device tensors (e.g. the
        output of utils.prepare_targets, which then never leaves the GPU)"""
        M = self.logits_view.buf.M
        if torch.is_tensor(y):
            yt = y.reshape(-1).to(torch.float32)
        else:
            yt = torch.from_numpy(np.ascontiguousarray(np.asarray(y, np.float32).reshape(-1)))
        assert yt.numel() == M, (yt.numel(), M)
        self.labels.copy_(yt.to(self.device, non_blocking=True))
        if sw is None:
            # Keras without sample weights: plain mean over all B*HW pixels; void rows contribute zero loss and zero
            # gradient through the one-hot (utils.py:129), not through a weight
            capi.call("dl3_fill", ptr(self.sweights), 1.0, M, torch.cuda.current_stream().cuda_stream)
        elif torch.is_tensor(sw):
            self.sweights.copy_(sw.reshape(-1).to(self.device, torch.float32, non_blocking=True))
        else:
            self.sweights.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(sw, np.float32).reshape(-1))).to(self.device))

    def count_nnz(self):
        """count(w != 0) of the resident sample weights (device count, one scalar read back)"""
        tmp = torch.zeros(4, dtype=torch.float32, device=self.device)
        capi.call("dl3_count_nonzero", ptr(self.sweights), self.logits_view.buf.M, ptr(tmp),
                  torch.cuda.current_stream().cuda_stream)
        return float(tmp[0].item())

    def set_nnz(self, value):
        """external_nnz engines: the FIXED normaliser the loss kernel divides by (default: pixels per image); the
        optimizer multiplies it back and divides by the global count (train_step).  Tests use it to give engines of
        different batch sizes one common normaliser."""
        assert self.external_nnz
        self.nnz_host = float(value)
        capi.call("dl3_fill", ptr(self.nnz), float(value), 1, torch.cuda.current_stream().cuda_stream)

    def loss_value(self):
        """the loss of the last step as a float (one device read): sum(l*w) / count(w != 0) — over the GLOBAL batch on a
        data-parallel engine, from the arena tail the all-reduce summed"""
        if not self.external_nnz:
            return float(self.loss[0].item())
        t = self.grads[self.tail:self.tail + 2].cpu()
        return float(t[1]) * self.nnz_host / max(float(t[0]), 1e-20)

    def loss_handle(self):
        """the same without the read: a LazyLoss that fetches when (if) somebody looks at it"""
        if not self.external_nnz:
            return LazyLoss(self.loss[0:1].clone(), None, 1.0)
        t = self.grads[self.tail:self.tail + 2].clone()
        return LazyLoss(t[1:2], t[0:1], self.nnz_host)

    def seg_counts(self, y):
        """after forward(): per-image, per-class pixel counts [B,3,C] of the argmax mask against labels y
        (dl3_argmax + dl3_seg_counts on the device; utils.Jaccard_from_counts turns them into the metric)"""
        v = self.logits_view
        st = torch.cuda.current_stream().cuda_stream
        pred = torch.empty(v.buf.M, dtype=torch.int32, device=self.device)
        capi.call("dl3_argmax", ptr(v.buf.t), pred.data_ptr(), v.buf.M, v.C, st)
        if torch.is_tensor(y):
            yt = y.reshape(-1).to(self.device, torch.float32)
        else:
            yt = torch.from_numpy(np.ascontiguousarray(np.asarray(y, np.float32).reshape(-1))).to(self.device)
        assert yt.numel() == v.buf.M, (yt.numel(), v.buf.M)
        counts = torch.empty(self.B, 3, v.C, dtype=torch.int32, device=self.device)
        capi.call("dl3_seg_counts", pred.data_ptr(), yt.data_ptr(), self.B, v.buf.M // self.B, v.C, counts.data_ptr(), st)
        return counts.cpu().numpy()

    def fwd_bwd(self):
        """forward + loss + backward on the resident batch (the benchmarked hot path)"""
        self._prep()
        if self.use_graph and self._calls >= 1:
            if self.graph is None:
                # the first call ran eagerly (module loading, allocator warm-up); capture one replayable hipGraph now
                torch.cuda.synchronize()
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.run_ops(self.ops_fwd)
                        self._run_bwd()
                    self.graph = g
                except Exception as e:  # pragma: no cover - depends on the runtime
                    print("dl3: hipGraph capture failed (%s); running eagerly" % e)
                    self.use_graph = False
                    torch.cuda.synchronize()
                    return self.fwd_bwd()
            self.graph.replay()
        else:
            self.run_ops(self.ops_fwd)
            self._run_bwd()
        self._calls += 1

    def _run_bwd(self):
        if self.fork and self._side:
            self.run_ops_forked(self.ops_bwd)
        else:
            self.run_ops(self.ops_bwd)

    def adam(self, opt=None, grad_scale=1.0, norm=None):
        """Keras Adam with decay (notebook cell 2): lr_t = lr/(1+decay*it) * sqrt(1-b2^t)/(1-b1^t).
        norm (the default on data-parallel / external_nnz engines, and the only valid choice there): the gradient scale is
        finished on the device, normaliser / count_all from the arena tail (dl3_adam_step_norm); grad_scale is ignored.
        An external_nnz engine's loss kernel divides by the FIXED c0, so adam(opt, scale) without norm would silently
        apply grad * scale / c0 instead of grad / count_all: refused."""
        if norm is None:
            norm = self.external_nnz
        if self.external_nnz and not norm:
            raise ValueError("Engine.adam(norm=False) on an external_nnz (data-parallel) engine: its gradients are "
                             "normalised by a fixed constant and must be finished by dl3_adam_step_norm")
        o = dict(lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6)
        o.update(opt or {})
        it = self.iteration
        t = it + 1
        lr = o["lr"] / (1.0 + o["decay"] * it)
        lr_t = lr * math.sqrt(1.0 - o["beta_2"] ** t) / (1.0 - o["beta_1"] ** t)
        if norm:
            assert self.external_nnz
            capi.call("dl3_adam_step_norm", ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                      self.n_param, lr_t, o["beta_1"], o["beta_2"], o["epsilon"], self.nnz_host,
                      self.grads.data_ptr() + 4 * self.tail, torch.cuda.current_stream().cuda_stream)
        else:
            capi.call("dl3_adam_step", ptr(self.params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v),
                      self.n_param, lr_t, o["beta_1"], o["beta_2"], o["epsilon"], grad_scale,
                      torch.cuda.current_stream().cuda_stream)
        self.iteration += 1
        self.dirty = True

    def adopt_optimizer_state(self, other):
        """continue another training engine's optimizer (same model, other batch size): Adam moments, iteration count
        and the dropout step move device-to-device; the flat layouts are identical by construction"""
        if other is self or not (self.training and other.training):
            return
        if other.adam_m.numel() != self.adam_m.numel() or other.n_param != self.n_param:
            raise RuntimeError("optimizer state layouts differ (engines of two different compile() calls?)")
        self.adam_m.copy_(other.adam_m)
        self.adam_v.copy_(other.adam_v)
        self.drop_step.copy_(other.drop_step)
        self.iteration = other.iteration

    def train_step(self, x, y, sw=None, opt=None, comm=None, lazy=False):
        """one step on (x, y, sw).  Data parallel (external_nnz): ONE loss over the global batch,
        L = sum_all(l*w) / count_all(w != 0) — every rank differentiates sum_shard(l*w) / c0, the all-reduce sums
        gradients, counts and loss sums, Adam applies c0 / count_all on the device: nothing in front of the captured
        step, one collective behind it, no host round trip.  lazy: the loss comes back as a LazyLoss (no device read)."""
        self.set_input(x)
        self.set_targets(y, sw)
        self.fwd_bwd()
        if self.external_nnz:
            if comm is not None:
                comm.allreduce_grads(self.grads)
            self.adam(opt, norm=True)
        else:
            scale = 1.0
            if comm is not None:
                scale = comm.allreduce_grads(self.grads)
            self.adam(opt, scale)
        return self.loss_handle() if lazy else self.loss_value()

    def grad_of(self, name):
        kind, off, n, shp, _ = self.slots[name]
        assert kind == "p"
        return self._unpad(name, self.grads[off:off + n].cpu().numpy().reshape(shp))


class LazyLoss:
    """the loss of one step, still on the device: num [* scale / den], read (one synchronising copy) only when somebody
    converts it — Model.fit / fit_generator collect these and read them once per epoch instead of stalling every step"""

    def __init__(self, num, den, scale):
        self._num, self._den, self._scale, self._value = num, den, scale, None

    def __float__(self):
        if self._value is None:
            n = float(self._num.cpu()[0])
            self._value = n if self._den is None else n * self._scale / max(float(self._den.cpu()[0]), 1e-20)
            self._num = self._den = None
        return self._value

    def __repr__(self):
        return repr(float(self))

    def __format__(self, spec):
        return format(float(self), spec)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(float(self), dtype=dtype)


# ======================================================================================
# units
# ======================================================================================


class _ConvBase:
    def __init__(self, eng, layer, inv, outv, want_stat):
        self.eng, self.layer, self.inv, self.outv = eng, layer, inv, outv
        self.bn = None
        self.stat, self.P = None, 0
        self.want_stat = want_stat
        eng._consume(inv)

    def wname(self):
        return self.layer.name + "/kernel:0"


class PwUnit(_ConvBase):
    """Conv2D 1x1 (+bias): dl3_pwconv_fwd / _bwd_weight / _bwd_data.
    wrow0: first row of the layer's [K_total][N] kernel this unit multiplies with (its K = inv.C rows from there);
    img_add: Buf [B,1,1,N] added per image (dl3_pwconv_fwd_add) — together they split a convolution over a Concatenate
    whose first input is the broadcast of a 1x1 map (the ASPP image-pooling branch) into a per-image GEMM of B rows and
    a per-pixel GEMM over the remaining channels (Engine._lo_Conv2D)."""

    def __init__(self, eng, layer, inv, outv, want_stat, wrow0=0, img_add=None, use_bias=True):
        super().__init__(eng, layer, inv, outv, want_stat)
        self.M, self.K, self.N = inv.buf.M, inv.C, outv.C
        self.wrow0, self.img_add = int(wrow0), img_add
        self.bias = layer.name + "/bias:0" if (layer.cfg["use_bias"] and use_bias) else None
        if want_stat:
            self.P = eng.lib.dl3_pwconv_partials(self.M, self.K, self.N)
            self.stat = eng.empty(self.P * self.N * 2)
        s, t, a = inv.xform()
        w = eng.wptr(self.wname()) + 4 * self.wrow0 * self.N
        if img_add is None and not want_stat and inv.buf.H == 1 and inv.buf.W == 1:
            # one row per image (the ASPP image-pooling branch and its share of concat_projection): accumulated in double
            # (dl3_pwconv_fwd_rows) — the result is added to every pixel of the map, its rounding error is coherent
            self.fwd_rec = eng.op(eng.ops_fwd, "dl3_pwconv_fwd_rows", inv.p(), inv.ld, s, t, a, w,
                                  eng.wptr(self.bias) if self.bias else None, outv.p(), outv.ld, self.M, self.K, self.N,
                                  None, 0, 1)
        elif img_add is None:
            self.fwd_rec = eng.op(eng.ops_fwd, "dl3_pwconv_fwd", inv.p(), inv.ld, s, t, a, w,
                                  eng.wptr(self.bias) if self.bias else None, outv.p(), outv.ld, self.M, self.K, self.N,
                                  ptr(self.stat))
        else:
            assert img_add.M == eng.B and img_add.ld == self.N and self.M % eng.B == 0
            eng._consume(View(img_add, 0, self.N))
            self.fwd_rec = eng.op(eng.ops_fwd, "dl3_pwconv_fwd_add", inv.p(), inv.ld, s, t, a, w,
                                  eng.wptr(self.bias) if self.bias else None, outv.p(), outv.ld, self.M, self.K, self.N,
                                  ptr(self.stat), ptr(img_add.t), self.N, self.M // eng.B)

    def set_output_offset(self, neg_offset_ptr):
        """donate the GEMM's free bias slot (a BatchNorm'ed Conv2D has use_bias=False) to -moving_mean: the output tensor
        then holds y - mean, and ONLY affine consumers (scale * stored + beta) may read it — the slice is recorded in
        Buf.centred and Engine._check_centred_consumers refuses a raw view of it"""
        # a convolution with a per-image term (concat_projection: deeplabv3p.py:402-406) subtracts the mean THERE — the
        # few-row GEMM accumulates in double, so addend - mean is rounded once and the per-pixel epilogue adds one number
        # instead of (acc - mean) + addend, whose first step cancels against a mean that contains the addend's average
        rec = self.fwd_rec
        iu = getattr(self, "img_unit", None)
        if iu is not None and iu.fwd_rec[0] == "dl3_pwconv_fwd_rows":
            rec = iu.fwd_rec
        assert rec[0] in ("dl3_pwconv_fwd", "dl3_pwconv_fwd_add", "dl3_pwconv_fwd_rows") and rec[2][6] is None and self.bias is None
        rec[2][6] = neg_offset_ptr
        self.outv.buf.centred.add(self.outv.off)

    def _bwd_img_add(self, g, ldg, y, ldy, cA, cB, cC):
        """gradient of the per-image addend: S[img, n] = sum over the image's pixels of dY[m, n], dY = cA*g + cB*y + cC
        (two per-image column sums with the coefficients folded in, then their sum)"""
        eng, N, B = self.eng, self.N, self.eng.B
        HW = self.M // B
        ab = self.img_add
        gout, add, last = eng.contrib_kernel(ab)
        assert add is None and last and not ab.bns
        if cA is None:
            eng.op(eng.ops_bwd, "dl3_gap_fwd", g, ldg, None, None, ACT_NONE, ptr(gout), B, HW, N, 1.0)
            return
        sg, sy, zero = eng.empty(B * N), eng.empty(B * N), eng.zeros(N)
        eng.op(eng.ops_bwd, "dl3_gap_fwd", g, ldg, cA, cC, ACT_NONE, ptr(sg), B, HW, N, 1.0)
        eng.op(eng.ops_bwd, "dl3_gap_fwd", y, ldy, cB, ptr(zero), ACT_NONE, ptr(sy), B, HW, N, 1.0)
        eng.op(eng.ops_bwd, "dl3_affine_add", ptr(sg), N, None, None, ACT_NONE, ptr(sy), N, None, None, ACT_NONE,
               ptr(gout), N, B, N, 0.0, 0, None)

    def dy_mat_ok(self):
        """(decided ONCE per unit: the buffer sizing in _lower_backward and bwd() must agree)"""
        if getattr(self, "_dy_ok", None) is None:
            self._dy_ok = bool(self._dy_mat_rule())
        return self._dy_ok

    def _dy_mat_rule(self):
        """does the weight-gradient launch of this convolution also write dY for its bwd-data launch?  (its output is
        BatchNorm'ed — otherwise dY is g itself —, it has a weight gradient to compute and data gradient to hand on, the
        backward fork is off — the two launches must stay ordered on one stream — and the shape rule of Engine.dy_mat holds)"""
        eng, outv = self.eng, self.outv
        has_bn = any(off == outv.off for _, off, _ in outv.buf.bns)
        return (has_bn and eng.dy_mat and (self.N <= self.K or self.K >= eng.dy_mat_mink) and self.N % 4 == 0
                and not self.fused_ok()
                and not eng.fork and not self.bias and eng.trainable(self.wname()) and self.inv.buf.requires_grad)

    def fused_ok(self):
        if getattr(self, "_fused_ok", None) is None:
            self._fused_ok = bool(self._fused_rule())
        return self._fused_ok

    def _fused_rule(self):
        """both gradients in one pass (dl3_pwconv_bwd_fused): a small weight matrix, many pixel rows, a trainable kernel
        without bias, a data gradient to hand on, whole aligned tensors on both sides"""
        eng, inv, outv = self.eng, self.inv, self.outv
        if not (eng.fused_bwd and not eng.fork and self.img_add is None and not self.bias and self.wrow0 == 0
                and self.M >= eng.fused_min_rows and eng.trainable(self.wname()) and inv.buf.requires_grad
                and inv.ld % 4 == 0 and inv.off % 4 == 0 and outv.ld % 4 == 0 and outv.off % 4 == 0):
            return False
        sup = eng.lib.dl3_pwconv_bwd_fused_supported(self.M, self.K, self.N)
        # (2: any epilogue; 1 — K > 64 —: no residual addend and sums against the forward input only, i.e. an input with
        # one consumer that is not a bare residual Add output)
        ib = inv.buf
        return sup == 2 or (sup == 1 and ib.expected == 1 and (bool(ib.bns) or id(ib) not in eng.add_of_buf))

    def _bwd_fused(self):
        eng, inv, outv = self.eng, self.inv, self.outv
        M, K, N = self.M, self.K, self.N
        g, ldg, y, ldy, cA, cB, cC = eng.grad_operand(outv)
        s, t, a = inv.xform()
        ibuf = inv.buf
        wT = eng.empty(K * N)
        eng.transpose(eng.wptr(self.wname()), wT, K, N)
        gout, add, last = eng.contrib_kernel(ibuf)
        need_stat = last and bool(ibuf.bns)
        if need_stat and (inv.off != 0 or inv.C != ibuf.ld):
            raise NotImplementedError("BN-backward statistics through a channel slice")
        S = eng.lib.dl3_pwconv_bwd_fused_splits(M, K, N)
        ws = eng.lib.dl3_pwconv_bwd_fused_workspace(M, K, N)
        tgt = None
        if last and not need_stat and a == ACT_NONE and inv.off == 0 and inv.C == ibuf.ld:
            tgt = eng.alias_stats_target(ibuf)   # see bwd(): the sums of the BatchNorm the gradient reaches through the Add
        sbuf = tgt if tgt is not None else (ibuf if need_stat else None)
        dpart = eng.empty(S * sbuf.ld * 2) if sbuf is not None else None
        own = eng.empty(ws // 4 + 4)
        eng.own_fold_ws_bytes += ws + 16
        eng.op(eng.ops_bwd, "dl3_pwconv_bwd_fused", inv.p(), inv.ld, s, t, a, g, ldg, y, ldy, cA, cB, cC, ptr(wT), None,
               gout.data_ptr() + 4 * inv.off, ibuf.ld, (add.data_ptr() + 4 * inv.off) if add is not None else None, ibuf.ld,
               (sbuf.t.data_ptr() + (4 * inv.off if sbuf is ibuf else 0)) if sbuf is not None else None,
               sbuf.ld if sbuf is not None else 0, sbuf.vptr(V_MEAN, inv.off if sbuf is ibuf else 0) if sbuf is not None else None,
               sbuf.vptr(V_INVSTD, inv.off if sbuf is ibuf else 0) if sbuf is not None else None, ptr(dpart), M, K, N,
               ptr(own), ws)
        if eng.defer_fold(S, K * N):
            eng.fold(ptr(own), S, K * N, eng.gptr(self.wname()))
        else:
            eng.op(eng.ops_bwd, "dl3_reduce_partials", ptr(own), S, K * N, eng.gptr(self.wname()))
        if tgt is not None:
            eng.prestat[id(tgt)] = (dpart, S, tgt.ld)
        elif need_stat:
            eng.finish_bn_bwd(ibuf, dpart, S, ibuf.ld)

    def bwd(self):
        if self.fused_ok():
            return self._bwd_fused()
        eng, inv, outv = self.eng, self.inv, self.outv
        M, K, N = self.M, self.K, self.N
        g, ldg, y, ldy, cA, cB, cC = eng.grad_operand(outv)
        s, t, a = inv.xform()
        wsrc = eng.wptr(self.wname()) + 4 * self.wrow0 * N
        dy = ptr(eng.dy_buf) if (cA is not None and self.dy_mat_ok()) else None
        if eng.trainable(self.wname()) or (self.bias and eng.trainable(self.bias)):
            ws = eng.lib.dl3_pwconv_bwd_weight_workspace(M, K, N)
            S = eng.lib.dl3_pwconv_bwd_weight_splits(M, K, N, 1 if cA else 0)
            fn, tail = ("dl3_pwconv_bwd_weight_dy", (dy, N)) if dy else ("dl3_pwconv_bwd_weight", ())
            if eng.defer_fold(S, K * N) and not self.bias and eng.trainable(self.wname()):
                # the launch leaves its [S][K][N] slabs in a workspace of its own; folded at the end of the pass
                own = eng.empty(ws // 4 + 4)
                eng.own_fold_ws_bytes += ws + 16
                eng.op_side(eng.ops_bwd, fn, inv.p(), inv.ld, s, t, a, g, ldg, y, ldy, cA, cB, cC,
                            None, None, M, K, N, ptr(own), ws, *tail)
                eng.fold(ptr(own), S, K * N, eng.gptr(self.wname()) + 4 * self.wrow0 * N)
            else:
                eng.op_ws_side(eng.ops_bwd, fn, ws, 17, inv.p(), inv.ld, s, t, a, g, ldg, y, ldy, cA, cB,
                               cC, eng.gptr(self.wname()) + (4 * self.wrow0 * N if eng.trainable(self.wname()) else 0),
                               eng.gptr(self.bias) if self.bias else None, M, K, N, 0, ws, *tail)
        if dy:
            # from here on the layer's gradient operand is the one tensor the weight-gradient launch just wrote
            g, ldg, y, ldy, cA, cB, cC = dy, N, None, 0, None, None, None
        if self.img_add is not None:
            self._bwd_img_add(g, ldg, y, ldy, cA, cB, cC)
        ibuf = inv.buf
        if not ibuf.requires_grad:
            return
        wT = eng.empty(K * N)
        eng.transpose(wsrc, wT, K, N)
        gout, add, last = eng.contrib_kernel(ibuf)
        need_stat = last and bool(ibuf.bns)
        P = eng.lib.dl3_pwconv_partials(M, N, K)
        dpart = eng.empty(P * ibuf.ld * 2) if need_stat else None
        if need_stat and (inv.off != 0 or inv.C != ibuf.ld):
            raise NotImplementedError("BN-backward statistics through a channel slice")
        need_x = a != ACT_NONE or need_stat
        tgt = None
        if last and not need_x and inv.off == 0 and inv.C == ibuf.ld:
            # the epilogue's forward-input slot is free (no mask, no BatchNorm on the Add output): use it for the sums of
            # the BatchNorm this gradient reaches unchanged through the residual Add
            tgt = eng.alias_stats_target(ibuf)
        if tgt is not None:
            dpart = eng.empty(P * tgt.ld * 2)
            eng.op(eng.ops_bwd, "dl3_pwconv_bwd_data", g, ldg, y, ldy, cA, cB, cC, ptr(wT), ptr(gout), ibuf.ld,
                   ptr(tgt.t), tgt.ld, None, None, ACT_NONE, ptr(add), ibuf.ld, 1, 1.0,
                   tgt.vptr(V_MEAN), tgt.vptr(V_INVSTD), ptr(dpart), M, K, N)
            eng.prestat[id(tgt)] = (dpart, P, tgt.ld)
            return
        eng.op(eng.ops_bwd, "dl3_pwconv_bwd_data", g, ldg, y, ldy, cA, cB, cC, ptr(wT),
               gout.data_ptr() + 4 * inv.off, ibuf.ld, inv.p() if need_x else None, inv.ld,
               s if a != ACT_NONE else None, t if a != ACT_NONE else None, a,
               (add.data_ptr() + 4 * inv.off) if add is not None else None, ibuf.ld, 1, 1.0,
               ibuf.vptr(V_MEAN, inv.off) if need_stat else None, ibuf.vptr(V_INVSTD, inv.off) if need_stat else None,
               ptr(dpart), M, K, N)
        if need_stat:
            eng.finish_bn_bwd(ibuf, dpart, P, ibuf.ld)


class DwUnit(_ConvBase):
    """DepthwiseConv2D 3x3: dl3_dwconv3x3_fwd / dl3_dwconv3x3_bwd (fused data+weight gradient)"""

    def __init__(self, eng, layer, inv, outv, stride, rate, pt, pl, want_stat):
        super().__init__(eng, layer, inv, outv, want_stat)
        N, H, W, C = inv.shape
        Ho, Wo = outv.buf.H, outv.buf.W
        self.geom = (eng.B, H, W, C, stride, rate, pt, pl, Ho, Wo)
        self.P = eng.lib.dl3_dwconv3x3_partials(eng.B, H, W, C, stride, rate, Ho, Wo, eng.dw_impl)
        if want_stat:
            self.stat = eng.empty(self.P * C * 2)
        s, t, a = inv.xform()
        eng.op(eng.ops_fwd, "dl3_dwconv3x3_fwd", inv.p(), s, t, a, eng.wptr(self.wname()), outv.p(), *self.geom,
               ptr(self.stat), eng.dw_impl)

    def wname(self):
        return self.layer.name + "/depthwise_kernel:0"

    def bwd(self):
        eng, inv, outv = self.eng, self.inv, self.outv
        C = inv.C
        g, ldg, y, ldy, cA, cB, cC = eng.grad_operand(outv)
        assert ldg == C
        s, t, a = inv.xform()
        ibuf = inv.buf
        wpart = eng.empty(self.P * 9 * C)
        gout = add = dpart = None
        need_stat = False
        tgt = None
        if ibuf.requires_grad:
            gout, add, last = eng.contrib_kernel(ibuf)
            need_stat = last and bool(ibuf.bns)
            dpart = eng.empty(self.P * C * 2) if need_stat else None
            B_, H, W, _, stride, rate, pt, pl, Ho, Wo = self.geom
            if (last and not ibuf.bns and stride == 1 and Ho == H and Wo == W and pt == rate and pl == rate
                    and eng.dw_impl in (IMPL_AUTO, capi.IMPL_MARCH)):
                # this launch completes the gradient of a residual Add's output (Xception's `sum` shortcuts): the sums of the
                # BatchNorm it reaches unchanged through the Add ride along (round 4; PwUnit.bwd does the same for
                # MobileNetV2's blocks) instead of a dl3_grad_finish pass over g and y
                tgt = eng.alias_stats_target(ibuf)
        if tgt is not None:
            dpart = eng.empty(self.P * C * 2)
            eng.op(eng.ops_bwd, "dl3_dwconv3x3_bwd_sx", g, y, cA, cB, cC, inv.p(), s, t, a, eng.wptr(self.wname()),
                   ptr(gout), ptr(add), ptr(tgt.t), tgt.vptr(V_MEAN), tgt.vptr(V_INVSTD), ptr(dpart), ptr(wpart),
                   *self.geom, eng.dw_impl)
            eng.prestat[id(tgt)] = (dpart, self.P, C)
        else:
            eng.op(eng.ops_bwd, "dl3_dwconv3x3_bwd", g, y, cA, cB, cC, inv.p(), s, t, a, eng.wptr(self.wname()),
                   ptr(gout), ptr(add), ibuf.vptr(V_MEAN) if need_stat else None,
                   ibuf.vptr(V_INVSTD) if need_stat else None, ptr(dpart), ptr(wpart), *self.geom, eng.dw_impl)
        eng.fold(ptr(wpart), self.P, 9 * C, eng.gptr(self.wname()))
        if need_stat:
            eng.finish_bn_bwd(ibuf, dpart, self.P, C)


class Conv3Unit(_ConvBase):
    """dense Conv2D 3x3 (stem)"""

    def __init__(self, eng, layer, inv, outv, stride, pt, pl, want_stat):
        super().__init__(eng, layer, inv, outv, want_stat)
        N, H, W, Cin = inv.shape
        Ho, Wo, Cout = outv.buf.H, outv.buf.W, outv.C
        if inv.off != 0 or inv.ld != Cin or outv.off != 0 or outv.ld != Cout:
            raise NotImplementedError("dense 3x3 conv on channel slices")
        self.geom = (eng.B, H, W, Cin, Cout, stride, pt, pl, Ho, Wo)
        # 32/64-channel tensors (xception entry_flow_conv1_2, 32 -> 64): the im2col-free matrix-pipe kernels (taps
        # gathered straight into the MFMA operand); the 3-channel stem convs have their own MFMA kernels behind
        # dl3_conv3x3_fwd
        self.gemm = bool(eng.lib.dl3_conv3x3_mfma_supported(Cin, Cout))
        s, t, a = inv.xform()
        if self.gemm:
            self.P = eng.lib.dl3_conv3x3_mfma_partials(eng.B, Ho, Wo)
            if want_stat:
                self.stat = eng.empty(self.P * Cout * 2)
            eng.op(eng.ops_fwd, "dl3_conv3x3_mfma_fwd", inv.p(), s, t, a, eng.wptr(self.wname()), outv.p(), *self.geom,
                   ptr(self.stat))
            return
        self.P = eng.lib.dl3_conv3x3_partials(eng.B, Ho, Wo, Cout)
        if want_stat:
            self.stat = eng.empty(self.P * Cout * 2)
        eng.op(eng.ops_fwd, "dl3_conv3x3_fwd", inv.p(), s, t, a, eng.wptr(self.wname()), outv.p(), *self.geom,
               ptr(self.stat))

    def _bwd_gemm(self):
        eng, inv, outv = self.eng, self.inv, self.outv
        B, H, W, Cin, Cout, stride, pt, pl, Ho, Wo = self.geom
        g, ldg, y, ldy, cA, cB, cC = eng.grad_operand(outv)
        assert ldg == Cout and (y is None or ldy == Cout)
        s, t, a = inv.xform()
        if eng.trainable(self.wname()):
            ws = eng.lib.dl3_conv3x3_mfma_bwd_weight_workspace(*self.geom[:6], Ho, Wo)
            eng.op_ws(eng.ops_bwd, "dl3_conv3x3_mfma_bwd_weight", ws, 20, inv.p(), s, t, a, g, y, cA, cB, cC,
                      eng.gptr(self.wname()), *self.geom, 0, ws)
        ibuf = inv.buf
        if not ibuf.requires_grad:
            return
        wT = eng.empty(9 * Cin * Cout)
        eng.transpose(eng.wptr(self.wname()), wT, 9 * Cin, Cout)
        gout, add, last = eng.contrib_kernel(ibuf)
        need_stat = last and bool(ibuf.bns)
        P = eng.lib.dl3_conv3x3_mfma_partials(B, H, W)
        dpart = eng.empty(P * Cin * 2) if need_stat else None
        need_x = a != ACT_NONE or need_stat
        eng.op(eng.ops_bwd, "dl3_conv3x3_mfma_bwd_data", g, y, cA, cB, cC, ptr(wT), ptr(gout),
               inv.p() if need_x else None, s if a != ACT_NONE else None, t if a != ACT_NONE else None, a, ptr(add),
               ibuf.vptr(V_MEAN) if need_stat else None, ibuf.vptr(V_INVSTD) if need_stat else None, ptr(dpart),
               *self.geom)
        if need_stat:
            eng.finish_bn_bwd(ibuf, dpart, P, Cin)

    def bwd(self):
        if self.gemm:
            return self._bwd_gemm()
        eng, inv, outv = self.eng, self.inv, self.outv
        B, H, W, Cin, Cout, stride, pt, pl, Ho, Wo = self.geom
        g, ldg, y, ldy, cA, cB, cC = eng.grad_operand(outv)
        s, t, a = inv.xform()
        if eng.trainable(self.wname()):
            wpart = eng.empty(self.P * 9 * Cin * Cout)
            eng.op(eng.ops_bwd, "dl3_conv3x3_bwd_weight", inv.p(), s, t, a, g, y, cA, cB, cC, ptr(wpart), *self.geom)
            eng.fold(ptr(wpart), self.P, 9 * Cin * Cout, eng.gptr(self.wname()))
        ibuf = inv.buf
        if not ibuf.requires_grad:
            return
        gout, add, last = eng.contrib_kernel(ibuf)
        need_stat = last and bool(ibuf.bns)
        P = eng.lib.dl3_conv3x3_partials(B, H, W, Cin)
        dpart = eng.empty(P * Cin * 2) if need_stat else None
        need_x = a != ACT_NONE or need_stat
        eng.op(eng.ops_bwd, "dl3_conv3x3_bwd_data", g, y, cA, cB, cC, eng.wptr(self.wname()), ptr(gout),
               inv.p() if need_x else None, s if a != ACT_NONE else None, t if a != ACT_NONE else None, a, ptr(add),
               ibuf.vptr(V_MEAN) if need_stat else None, ibuf.vptr(V_INVSTD) if need_stat else None, ptr(dpart),
               *self.geom)
        if need_stat:
            eng.finish_bn_bwd(ibuf, dpart, P, Cin)


class AddUnit:
    """Add (deeplabv3p.py:147-149,:201): out = T(a) + T(b); backward = gradient aliases"""

    def __init__(self, eng, a, b, outv):
        self.eng, self.a, self.b, self.outv = eng, a, b, outv
        eng._consume(a)
        eng._consume(b)
        sa, ta, aa = a.xform()
        sb, tb, ab = b.xform()
        eng.op(eng.ops_fwd, "dl3_affine_add", a.p(), a.ld, sa, ta, aa, b.p(), b.ld, sb, tb, ab, outv.p(), outv.ld,
               outv.buf.M, outv.C, 0.0, 0, None)

    def bwd(self):
        g = self.outv.buf.grad
        for v in (self.a, self.b):
            self.eng.contrib_passthrough(v.buf, v, g)


class MaterializeUnit:
    """Dropout(0.1) in training mode (deeplabv3p.py:410): out = T(x) * keepmask/(1-rate)"""

    def __init__(self, eng, inv, outv, rate, seed):
        self.eng, self.inv, self.outv, self.rate, self.seed = eng, inv, outv, rate, seed
        eng._consume(inv)
        s, t, a = inv.xform()
        eng.op(eng.ops_fwd, "dl3_affine_add", inv.p(), inv.ld, s, t, a, None, 0, None, None, ACT_NONE, outv.p(),
               outv.ld, outv.buf.M, outv.C, rate, seed, eng.drop_step.data_ptr())

    def bwd(self):
        g = self.outv.buf.grad
        self.eng.contrib_elementwise(self.inv.buf, self.inv, ptr(g), self.outv.ld, drop=(self.rate, self.seed))


class GapUnit:
    """global AveragePooling2D (deeplabv3p.py:375)"""

    def __init__(self, eng, inv, outv):
        self.eng, self.inv, self.outv = eng, inv, outv
        eng._consume(inv)
        s, t, a = inv.xform()
        self.HW = inv.buf.H * inv.buf.W
        eng.op(eng.ops_fwd, "dl3_gap_fwd", inv.p(), inv.ld, s, t, a, outv.p(), eng.B, self.HW, inv.C, 1.0 / self.HW)

    def bwd(self):
        g = self.outv.buf.grad
        if not self.inv.buf.requires_grad:
            return
        self.eng.contrib_elementwise(self.inv.buf, self.inv, ptr(g), self.outv.ld, gin_div=self.HW,
                                     gin_scale=1.0 / self.HW)


class ResizeUnit:
    """Lambda(tf.image.resize_bilinear) (deeplabv3p.py:382,:418,:439; utils.py:190)"""

    def __init__(self, eng, inv, outv, Ho, Wo):
        self.eng, self.inv, self.outv = eng, inv, outv
        eng._consume(inv)
        s, t, a = inv.xform()
        self.dims = (eng.B, inv.buf.H, inv.buf.W, Ho, Wo, inv.C)
        eng.op(eng.ops_fwd, "dl3_resize_bilinear_fwd", inv.p(), inv.ld, s, t, a, outv.p(), outv.ld, *self.dims)

    def bwd(self):
        eng, inv, outv = self.eng, self.inv, self.outv
        ibuf = inv.buf
        if not ibuf.requires_grad:
            return
        B, Hi, Wi, Ho, Wo, C = self.dims
        xfold = getattr(self, "xfold", None)
        if xfold is not None:
            # fused training tail: the loss kernel already folded the output columns (dl3_upsample_softmax_xent_fold)
            gout, add, last = eng.contrib_kernel(ibuf)
            eng.op(eng.ops_bwd, "dl3_resize_bilinear_bwd_rows", ptr(xfold), ptr(gout), ibuf.ld, B, Hi, Wi, Ho, C,
                   1 if add is not None else 0)
            if add is not None and add is not gout:
                raise NotImplementedError("resize backward with a foreign addend")
            return
        g = outv.buf.grad.data_ptr() + 4 * outv.off
        plain = inv.act == ACT_NONE and not ibuf.bns and inv.off == 0 and inv.C == ibuf.ld
        if Hi == 1 and Wi == 1:
            # a 1x1 source is a broadcast: its transpose is the per-image column sum
            tmp = eng.empty(B * C)
            eng.op(eng.ops_bwd, "dl3_gap_fwd", g, outv.ld, None, None, ACT_NONE, ptr(tmp), B, Ho * Wo, C, 1.0)
            eng.contrib_elementwise(ibuf, inv, ptr(tmp), C)
        elif plain:
            gout, add, last = eng.contrib_kernel(ibuf)
            ws = eng.lib.dl3_resize_bilinear_bwd_workspace(B, Hi, Wi, Ho, Wo, C)
            eng.op_ws(eng.ops_bwd, "dl3_resize_bilinear_bwd", ws, 11, g, outv.ld, ptr(gout), ibuf.ld, B, Hi, Wi, Ho, Wo,
                      C, 1 if add is not None else 0, 0, ws)
            if add is not None and add is not gout:
                raise NotImplementedError("resize backward with a foreign addend")
        else:
            tmp = eng.empty(ibuf.M * C)
            ws = eng.lib.dl3_resize_bilinear_bwd_workspace(B, Hi, Wi, Ho, Wo, C)
            eng.op_ws(eng.ops_bwd, "dl3_resize_bilinear_bwd", ws, 11, g, outv.ld, ptr(tmp), C, B, Hi, Wi, Ho, Wo, C, 0,
                      0, ws)
            eng.contrib_elementwise(ibuf, inv, ptr(tmp), C)


class ShuffleUnit:
    """Subpixel._phase_shift (subpixel.py:77-88)"""

    def __init__(self, eng, inv, outv, r, co):
        self.eng, self.inv, self.outv, self.r, self.co = eng, inv, outv, r, co
        eng._consume(inv)
        eng.op(eng.ops_fwd, "dl3_phase_shift", inv.p(), outv.p(), eng.B, inv.buf.H, inv.buf.W, co, r, 0)

    fused = False  # training: the loss kernel wrote the gradient in the unshuffled layout already (Engine._lo_softmax)

    def bwd(self):
        eng, inv = self.eng, self.inv
        if self.fused:
            assert inv.buf.grad is not None and not inv.buf.bns
            inv.buf.done += 1
            return
        gout, add, last = eng.contrib_kernel(inv.buf)
        assert add is None and not inv.buf.bns
        eng.op(eng.ops_bwd, "dl3_phase_shift", ptr(self.outv.buf.grad), ptr(gout), eng.B, inv.buf.H, inv.buf.W,
               self.co, self.r, 1)


class TapsUnit:
    """k x k taps of T(x) side by side in front of the GEMM of a Subpixel with kernel_size > 1 (subpixel.py:42-58)"""

    def __init__(self, eng, inv, outv, k, pt, pl):
        self.eng, self.inv, self.outv = eng, inv, outv
        eng._consume(inv)
        s, t, a = inv.xform()
        self.dims = (eng.B, inv.buf.H, inv.buf.W, inv.C, k, pt, pl, outv.buf.H, outv.buf.W)
        eng.op(eng.ops_fwd, "dl3_conv_taps_fwd", inv.p(), inv.ld, s, t, a, outv.p(), *self.dims)

    def bwd(self):
        eng, inv = self.eng, self.inv
        if not inv.buf.requires_grad:
            return
        tmp = eng.empty(inv.buf.M * inv.C)
        eng.op(eng.ops_bwd, "dl3_conv_taps_bwd", ptr(self.outv.buf.grad), ptr(tmp), *self.dims)
        eng.contrib_elementwise(inv.buf, inv, ptr(tmp), inv.C)


class SubsampleUnit:
    """row/column sampling in front of a strided 1x1 convolution (Xception shortcuts, deeplabv3p.py:143-145)"""

    def __init__(self, eng, inv, outv, stride):
        self.eng, self.inv, self.outv, self.stride = eng, inv, outv, stride
        eng._consume(inv)
        s, t, a = inv.xform()
        self.dims = (eng.B, inv.buf.H, inv.buf.W, inv.C, stride, outv.buf.H, outv.buf.W)
        eng.op(eng.ops_fwd, "dl3_subsample_fwd", inv.p(), inv.ld, s, t, a, outv.p(), *self.dims)

    def bwd(self):
        eng, inv = self.eng, self.inv
        if not inv.buf.requires_grad:
            return
        tmp = eng.empty(inv.buf.M * inv.C)
        eng.op(eng.ops_bwd, "dl3_subsample_bwd", ptr(self.outv.buf.grad), ptr(tmp), *self.dims)
        eng.contrib_elementwise(inv.buf, inv, ptr(tmp), inv.C)
