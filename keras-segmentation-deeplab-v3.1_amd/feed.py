"""Feeding the resident step from the host: decoded uint8 images and raw label maps cross PCIe as BYTES on a copy stream
while the previous step runs, and become the engine's float input / (Y, SW) targets on the device.

Reference being replaced: `SegmentationGenerator.__getitem__` (utils.py:360-402) builds float32 X, Y and the 'balanced'
sample weights per batch on the host, and `fit_generator` (utils.py:231-241) hands them to Keras, which copies 3 float
tensors per step (335 MB at B=64).  Here a batch is 1 byte per pixel-channel + 1 byte per label (100 + 34 MB at B=128):

    host (pinned)  --copy stream-->  device slot (uint8)  --compute stream-->  engine.xbuf (float32, widened),
                                                                             engine.labels / .sweights (dl3_prepare_targets)

Two device slots: while step i runs from slot i % 2 (already widened into the engine's own buffers), batch i+1 lands in the
other one.  The compute stream waits for a slot's copy event before it reads it; the copy stream waits for the slot's
consume event before it overwrites it.  The host never waits for a step: its only wait is for the H2D copy a slot issued
two steps earlier (long done) before that slot's pinned buffer is rewritten, and pinned batches skip even that.
"""
import numpy as np
import torch

from . import capi
from .capi import ptr


class BatchFeeder:
    def __init__(self, eng, n_classes, label_dtype=np.uint8, slots=2):
        if not eng.training:
            raise ValueError("BatchFeeder feeds a training engine (images + label maps)")
        if label_dtype not in (np.uint8, np.int32):
            raise ValueError("label maps are uint8 (cv2.imread(path, 0), utils.py:314) or int32 (utils.py:371)")
        self.eng, self.C, self.slots = eng, int(n_classes), int(slots)
        self.M = eng.logits_view.buf.M                    # label pixels per batch
        self.nx = eng.xbuf.t.numel()                      # image bytes per batch
        self.ldtype = torch.uint8 if label_dtype == np.uint8 else torch.int32
        self.lcode = capi.LABEL_U8 if label_dtype == np.uint8 else capi.LABEL_I32
        dev = eng.device
        self.dx = [torch.empty(self.nx, dtype=torch.uint8, device=dev) for _ in range(self.slots)]
        self.dl = [torch.empty(self.M, dtype=self.ldtype, device=dev) for _ in range(self.slots)]
        self.hx = [torch.empty(self.nx, dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
        self.hl = [torch.empty(self.M, dtype=self.ldtype).pin_memory() for _ in range(self.slots)]
        self.hist = torch.empty(eng.B, self.C + 1, dtype=torch.int32, device=dev)
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event() for _ in range(self.slots)]     # the slot's H2D copy has landed
        self.free = [torch.cuda.Event() for _ in range(self.slots)]      # the slot has been consumed
        self._used = [False] * self.slots
        self.bytes_per_batch = self.nx + self.M * (1 if label_dtype == np.uint8 else 4)

    @staticmethod
    def _checked(a, like, what):
        """images must BE bytes and label maps integers: a float array (the reference generator's own X / Y tensors,
        utils.py:360-402 — already normalised or augmented) would be truncated or wrapped by the narrowing copy into the
        uint8 / int32 slot instead of rejected"""
        dt = a.dtype if torch.is_tensor(a) else np.asarray(a).dtype
        name = str(dt).replace("torch.", "")
        if what == "images":
            if name != "uint8":
                raise ValueError("device feed: images must be uint8 [B,H,W,3] (decoded pixels 0..255), not %s — hand float "
                                 "tensors to train_on_batch / fit_generator(device_feed=False) instead" % name)
            return
        if name not in ("uint8", "int8", "int16", "int32", "int64", "uint16", "uint32", "uint64"):
            raise ValueError("device feed: label maps must be integers (uint8 / int32 class ids, 255 = void), not %s" % name)
        if like.dtype == torch.uint8 and name != "uint8":
            raise ValueError("device feed: this feeder was built for uint8 label maps, got %s" % name)
        if name in ("int64", "uint32", "uint64"):   # wider than the int32 slot: range-checked before narrowing
            n = a.numel() if torch.is_tensor(a) else np.asarray(a).size
            lo, hi = (int(a.min()), int(a.max())) if n else (0, 0)
            if lo < -2 ** 31 or hi >= 2 ** 31:
                raise ValueError("device feed: label values %d..%d do not fit the int32 slot" % (lo, hi))

    @classmethod
    def _pinned(cls, a, like, what="labels"):
        """a as a flat host tensor of like's dtype; pinned tensors pass through untouched (zero copy), anything else is
        copied into the slot's own pinned buffer"""
        cls._checked(a, like, what)
        if torch.is_tensor(a):
            t = a.reshape(-1)
            if t.is_pinned() and t.dtype == like.dtype:
                return t
            like.copy_(t.to(like.dtype))
            return like
        like.numpy()[...] = np.ascontiguousarray(a).reshape(-1)
        return like

    def stage(self, slot, images, labels):
        """enqueue batch (images uint8 [B,H,W,3], labels [B,H,W] or [B,HW]) for `slot` on the copy stream"""
        if self._used[slot]:
            self.ready[slot].synchronize()   # (the slot's previous H2D copy, two steps old, has left its pinned buffer)
        hx, hl = self._pinned(images, self.hx[slot], "images"), self._pinned(labels, self.hl[slot], "labels")
        assert hx.numel() == self.nx and hl.numel() == self.M, (hx.numel(), self.nx, hl.numel(), self.M)
        with torch.cuda.stream(self.copy_stream):
            if self._used[slot]:
                self.copy_stream.wait_event(self.free[slot])
            self.dx[slot].copy_(hx, non_blocking=True)
            self.dl[slot].copy_(hl, non_blocking=True)
            self.ready[slot].record(self.copy_stream)

    def consume(self, slot):
        """on the compute stream: slot -> the engine's resident input and targets (widening copy, dl3_prepare_targets)"""
        eng = self.eng
        st = torch.cuda.current_stream()
        st.wait_event(self.ready[slot])
        eng.xbuf.t.copy_(self.dx[slot])                                   # uint8 -> float32 on the device
        capi.call("dl3_prepare_targets", ptr(self.dl[slot]), self.lcode, eng.B, self.M // eng.B, self.C, ptr(eng.labels),
                  ptr(eng.sweights), ptr(self.hist), st.cuda_stream)
        self.free[slot].record(st)
        self._used[slot] = True

    def run(self, batches, step):
        """the pipelined loop: batches = iterable of (images, labels); step() = the resident step (fwd_bwd + Adam)"""
        it = iter(batches)
        nxt = next(it, None)
        if nxt is None:
            return 0
        self.stage(0, *nxt)
        i = 0
        while nxt is not None:
            cur = i % self.slots
            nxt = next(it, None)
            if nxt is not None:
                self.stage((i + 1) % self.slots, *nxt)
            self.consume(cur)
            step()
            i += 1
        return i
