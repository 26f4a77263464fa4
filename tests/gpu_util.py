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
