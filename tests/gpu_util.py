"""helpers for the -m gpu parity tests: device buffers + calling through the C ABI"""
import numpy as np
import torch

import dl3_amd  # noqa: F401
from dl3_amd import capi
from dl3_amd.capi import ptr


_KEEP = []  # device tensors must outlive the asynchronous launches that read them


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    _KEEP.append(t)
    return t


def empty(*shape):
    t = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
    _KEEP.append(t)
    return t


def release():
    torch.cuda.synchronize()
    del _KEEP[:]


def dropout_keep_mask(seed, n, rate, step=0):
    """host replica of dl3_uniform / dl3_step_seed (csrc/common.h): splitmix64 of (seed + step * odd constant,
    element index) -> keep mask"""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        seed = np.uint64(seed) + np.uint64(step) * np.uint64(0xD1B54A32D192ED03)
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(rate)).astype(np.float32)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    capi.call(name, *args, stream())


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def fold_partials(part, P, C):
    """[P][C][2] -> (sum0[C], sum1[C]) in float64"""
    p = host(part).reshape(P, C, 2).astype(np.float64)
    return p[:, :, 0].sum(0), p[:, :, 1].sum(0)


def np_act(x, act):
    if act == 1:
        return np.maximum(x, 0)
    if act == 2:
        return np.minimum(np.maximum(x, 0), 6)
    return x


def np_mask(z, act):
    if act == 1:
        return (z > 0).astype(z.dtype)
    if act == 2:
        return ((z > 0) & (z < 6)).astype(z.dtype)
    return np.ones_like(z)


# ---- tails (include/dl3.h dl3_tail): the finalize work done by a kernel's last-arriving workgroup ----------------
class TailCheck:
    """Builds a dl3_tail for an op test and checks what the kernel's tail wrote against the float64 finalize of the
    reference partial sums.  Tickets must be zero before the launch and zero again after it."""

    def __init__(self, kind, groups, C, rng, count, wsum_n=0):
        import ctypes
        self.kind, self.C, self.count = kind, C, float(count)
        self.ticket = torch.zeros(groups, dtype=torch.int32, device="cuda")
        self.gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
        self.beta = rng.normal(0, 1, C).astype(np.float32)
        self.mm = rng.normal(0, 1, C).astype(np.float32)
        self.mv = rng.uniform(0.5, 2, C).astype(np.float32)
        self.eps, self.mom = 1e-3, 0.99
        self.unb = self.count / max(self.count - 1, 1) * self.count / (self.count - (1 + self.eps))
        self.outs = [torch.full((C,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(4)]
        self.mmd, self.mvd = dev(self.mm), dev(self.mv)
        self.wsum = torch.full((max(wsum_n, 1),), float("nan"), dtype=torch.float32, device="cuda") if wsum_n else None
        t = capi.Tail()
        t.ticket = self.ticket.data_ptr()
        t.kind = kind
        t.batch_mode = 1
        t.eps, t.momentum, t.count, t.var_unbias = self.eps, self.mom, self.count, self.unb
        self._g, self._b = dev(self.gamma), dev(self.beta)
        t.gamma, t.beta = self._g.data_ptr(), self._b.data_ptr()
        self.struct = t
        if kind == capi.TAIL_BN_FWD:
            for i, o in enumerate(self.outs + [self.mmd, self.mvd]):
                t.o[i] = o.data_ptr()
        elif kind == capi.TAIL_BN_BWD:
            self.dg, self.db = empty(C), empty(C)
            for i, o in enumerate(self.outs[:3] + [self.dg, self.db]):
                t.o[i] = o.data_ptr()
        if self.wsum is not None:
            t.wsum = self.wsum.data_ptr()
        self.addr = ctypes.addressof(t)

    def set_bwd_inputs(self, mean, invstd):
        self._m, self._i = dev(mean), dev(invstd)
        self.mean_h, self.invstd_h = mean.astype(np.float64), invstd.astype(np.float64)
        self.struct.mean, self.struct.invstd = self._m.data_ptr(), self._i.data_ptr()

    def check_fwd(self, s1, s2, tol=1e-4):
        """s1 = sum(y), s2 = sum(y^2) per channel (float64 reference)"""
        assert int(host(self.ticket).max()) == 0 and int(host(self.ticket).min()) == 0, "tickets must return to zero"
        mean = s1 / self.count
        var = np.maximum(s2 / self.count - mean ** 2, 0)
        invstd = 1 / np.sqrt(var + self.eps)
        sc, sh, me, isd = [host(o).astype(np.float64) for o in self.outs]
        assert relerr(me, mean) < tol and relerr(isd, invstd) < tol
        assert relerr(sc, self.gamma * invstd) < tol and relerr(sh, self.beta - mean * self.gamma * invstd) < 10 * tol
        assert relerr(host(self.mmd), self.mom * self.mm + (1 - self.mom) * mean) < tol
        assert relerr(host(self.mvd), self.mom * self.mv + (1 - self.mom) * var * self.unb) < tol

    def check_bwd(self, d1, d2, tol=1e-3):
        """d1 = sum(g), d2 = sum(g * x_hat) per channel (float64 reference)"""
        assert int(host(self.ticket).max()) == 0 and int(host(self.ticket).min()) == 0, "tickets must return to zero"
        cA, cB, cC = [host(o).astype(np.float64) for o in self.outs[:3]]
        a = self.gamma * self.invstd_h
        b = -a * self.invstd_h * d2 / self.count
        assert relerr(host(self.dg), d2) < tol and relerr(host(self.db), d1) < tol
        assert relerr(cA, a) < 1e-5 and relerr(cB, b) < tol and relerr(cC, -a * d1 / self.count - b * self.mean_h) < tol
