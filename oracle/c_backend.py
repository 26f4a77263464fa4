"""C / OpenMP operators behind the numpy oracle's graph — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

oracle/c/dl3_ops.c restates the heavy operators of the path (DepthwiseConv2D, Conv2D, training-mode
BatchNormalization, legacy bilinear resize; forward and backward) in plain C loops.  This module wraps them with the
signatures and tape closures of their numpy twins in oracle/dl3_oracle.py, and `installed()` swaps them in:

    with c_backend.installed():
        loss, grads, logits, net = O.train_grads(params64, x, labels, weights, **kw)   # float64, 512x512 in seconds

so the float64 numpy oracle finishes at the sizes BASELINE.json quotes, where its pure-numpy operators take minutes.
The graph code (which layer follows which, names, epsilons, rates) stays dl3_oracle's; the operator arithmetic is a
third, independently written evaluation (numpy einsum/matmul vs torch/oneDNN vs these loops).  tests/test_oracle.py
checks the two operator sets against each other; bench.py times the float32 build as "CPU-A".

The shared library is built by `make -C oracle/c` (gcc -O3 -fopenmp; __graft_entry__.build() does it); nothing in the
product imports this module.
"""
import contextlib
import ctypes
import os
import subprocess

import numpy as np

from . import dl3_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(_HERE, "_build", "libdl3ops.so")
_lib = None


def build():
    subprocess.run(["make", "-C", os.path.join(_HERE, "c")], check=True, stdout=subprocess.DEVNULL)
    return LIBPATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            build()
        _lib = ctypes.CDLL(LIBPATH)
        _lib.dl3ops_max_threads.restype = ctypes.c_int
    return _lib


def set_threads(n):
    lib().dl3ops_set_threads(int(n))


def _sfx(a):
    if a.dtype == np.float64:
        return "f64", ctypes.c_double
    if a.dtype == np.float32:
        return "f32", ctypes.c_float
    raise TypeError("c_backend handles float32 / float64 arrays, not %s" % a.dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt=None):
    return np.ascontiguousarray(a, dtype=dt)


def _L(*v):
    return [ctypes.c_long(int(i)) for i in v]


def depthwise3x3(x, w, stride, rate, pad_t, pad_l, Ho, Wo, tape=None):
    """same contract as dl3_oracle.depthwise3x3 (deeplabv3p.py:73-74,:186-188)"""
    sfx, _ = _sfx(x)
    ins = (x, w)  # the tape links gradients to these very objects
    x, w = _c(x), _c(w, x.dtype)
    N, H, W, C = x.shape
    y = np.empty((N, Ho, Wo, C), x.dtype)
    geo = _L(N, H, W, C, stride, rate, pad_t, pad_l, Ho, Wo)
    getattr(lib(), "dl3ops_dw3x3_fwd_" + sfx)(_p(x), _p(w), _p(y), *geo)

    def bwd(g):
        g = _c(g, x.dtype)
        dx, dw = np.empty_like(x), np.empty_like(w)
        getattr(lib(), "dl3ops_dw3x3_bwd_" + sfx)(_p(x), _p(w), _p(g), _p(dx), _p(dw), *geo)
        return dx, dw

    return O._rec(tape, y, ins, bwd)


def conv2d(x, w, stride, pad_t, pad_l, Ho, Wo, bias=None, tape=None):
    """same contract as dl3_oracle.conv2d (deeplabv3p.py:99-116,:283,:318,:377,...)"""
    sfx, _ = _sfx(x)
    ins = (x, w) if bias is None else (x, w, bias)
    x, w = _c(x), _c(w, x.dtype)
    b = None if bias is None else _c(bias, x.dtype)
    N, H, W, Cin = x.shape
    k, Cout = w.shape[0], w.shape[3]
    y = np.empty((N, Ho, Wo, Cout), x.dtype)
    geo = _L(N, H, W, Cin, Cout, k, stride, pad_t, pad_l, Ho, Wo)
    getattr(lib(), "dl3ops_conv_fwd_" + sfx)(_p(x), _p(w), None if b is None else _p(b), _p(y), *geo)

    def bwd(g):
        g = _c(g, x.dtype)
        dx, dw = np.empty_like(x), np.empty_like(w)
        db = None if b is None else np.empty_like(b)
        getattr(lib(), "dl3ops_conv_bwd_" + sfx)(_p(x), _p(w), _p(g), _p(dx), _p(dw), None if db is None else _p(db), *geo)
        return (dx, dw) if b is None else (dx, dw, db)

    return O._rec(tape, y, ins, bwd)


_numpy_batchnorm = O.batchnorm


def batchnorm(x, gamma, beta, mmean, mvar, eps, training, momentum=0.99, tape=None, stats_out=None):
    """same contract as dl3_oracle.batchnorm; the training mode runs in C, inference stays numpy"""
    if not training:
        return _numpy_batchnorm(x, gamma, beta, mmean, mvar, eps, training, momentum, tape, stats_out)
    sfx, creal = _sfx(x)
    ins = (x, gamma, beta)
    x = _c(x)
    gm, bt = _c(gamma, x.dtype), _c(beta, x.dtype)
    C = x.shape[-1]
    M = x.size // C
    y, xhat = np.empty_like(x), np.empty_like(x)
    mean, var, invstd = np.empty(C), np.empty(C), np.empty(C)
    # (the epsilon floor of tf.nn.fused_batch_norm is applied here, on the graph side: the C operator takes what it is given)
    getattr(lib(), "dl3ops_bn_train_fwd_" + sfx)(_p(x), _p(gm), _p(bt), creal(O.fused_bn_epsilon(eps)), _p(y), _p(xhat), _p(mean), _p(var),
                                                 _p(invstd), *_L(M, C))
    if stats_out is not None:  # the moving-average convention is dl3_oracle's (Keras 2.2.4 on TF 1.13)
        unb = var * M / (M - 1) if M > 1 else var
        if M > 1.0 + eps:
            unb = unb * (M / (M - (1.0 + eps)))
        stats_out["mean"] = (momentum * mmean + (1 - momentum) * mean).astype(x.dtype)
        stats_out["var"] = (momentum * mvar + (1 - momentum) * unb).astype(x.dtype)
        stats_out["batch_mean"] = mean
        stats_out["batch_var"] = var

    def bwd(g):
        g = _c(g, x.dtype)
        dx = np.empty_like(x)
        dgamma, dbeta = np.empty(C), np.empty(C)
        getattr(lib(), "dl3ops_bn_train_bwd_" + sfx)(_p(g), _p(xhat), _p(gm), _p(invstd), _p(dx), _p(dgamma), _p(dbeta),
                                                     *_L(M, C))
        return dx, dgamma.astype(x.dtype), dbeta.astype(x.dtype)

    return O._rec(tape, y, ins, bwd)


def resize_bilinear_tf1(x, Ho, Wo, tape=None):
    """same contract as dl3_oracle.resize_bilinear_tf1 (deeplabv3p.py:382,:418,:439; utils.py:190); the source
    coordinates are computed in float32 as TF does (dl3_oracle._tf1_lerp), the interpolation runs in C"""
    sfx, _ = _sfx(x)
    ins = (x,)
    x = _c(x)
    N, Hi, Wi, C = x.shape
    ylo, yhi, wy = O._tf1_lerp(Ho, Hi)
    xlo, xhi, wx = O._tf1_lerp(Wo, Wi)
    tabs = [_c(ylo, np.int64), _c(yhi, np.int64), _c(wy, x.dtype), _c(xlo, np.int64), _c(xhi, np.int64), _c(wx, x.dtype)]
    y = np.empty((N, Ho, Wo, C), x.dtype)
    geo = _L(N, Hi, Wi, Ho, Wo, C)
    getattr(lib(), "dl3ops_resize_fwd_" + sfx)(_p(x), _p(y), *[_p(t) for t in tabs], *geo)

    def bwd(g):
        g = _c(g, x.dtype)
        dx = np.empty_like(x)
        getattr(lib(), "dl3ops_resize_bwd_" + sfx)(_p(g), _p(dx), *[_p(t) for t in tabs], *geo)
        return (dx,)

    return O._rec(tape, y, ins, bwd)


def _relu_like(x, hi, tape):
    sfx, creal = _sfx(x)
    ins = (x,)
    x = _c(x)
    y = np.empty_like(x)
    getattr(lib(), "dl3ops_relu_fwd_" + sfx)(_p(x), _p(y), creal(hi), ctypes.c_long(x.size))

    def bwd(g):
        g = _c(g, x.dtype)
        dx = np.empty_like(x)
        getattr(lib(), "dl3ops_relu_bwd_" + sfx)(_p(x), _p(g), _p(dx), creal(hi), ctypes.c_long(x.size))
        return (dx,)

    return O._rec(tape, y, ins, bwd)


def relu(x, tape=None):
    return _relu_like(x, 0.0, tape)


def relu6(x, tape=None):
    """relu(x, max_value=6.) (deeplabv3p.py:181,:192,:325)"""
    return _relu_like(x, 6.0, tape)


_OPS = ("depthwise3x3", "conv2d", "batchnorm", "resize_bilinear_tf1", "relu", "relu6")


@contextlib.contextmanager
def installed(threads=None):
    """run dl3_oracle's graph on the C operators inside the with-block"""
    lib()
    if threads:
        set_threads(threads)
    saved = {n: getattr(O, n) for n in _OPS}
    try:
        for n in _OPS:
            setattr(O, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(O, n, f)
