"""Subpixel layer + ICNR initialiser — drop-in for subpixel.py of the reference.

`Subpixel(filters, kernel_size, r, ...)` is a Conv2D with r*r*filters outputs followed by the
reference's own phase shift (subpixel.py:77-88), which is NOT tf.depth_to_space / torch.pixel_shuffle:
    out[n, ia*r+q, ib*r+p, ch] = conv[n, ia, ib, ch*r*r + p*r + q]
`ICNR` builds weights for the depth_to_space channel order (subpixel.py:27-39), i.e. the two are
mutually inconsistent in the reference; both quirks are reproduced exactly.
The phase shift runs in libdl3.so (dl3_phase_shift); ICNR is a host-side, once-per-model init.
"""
import numpy as np

from . import graph as G


class Subpixel(G.Conv2D):
    kind = "Subpixel"
    prefix = "subpixel"

    def __init__(self, filters, kernel_size, r, padding="valid", strides=(1, 1), activation=None, use_bias=True,
                 kernel_initializer="glorot_uniform", name=None, **kw):
        st = strides[0] if isinstance(strides, (tuple, list)) else strides
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        if int(st) != 1 or len(set(int(q) for q in ks)) != 1 or int(ks[0]) < 1:
            raise ValueError("Subpixel: square kernels with strides=1 only (got kernel_size=%r, strides=%r); the "
                             "reference never uses anything else (utils.py:195)" % (kernel_size, strides))
        if activation is not None:
            raise ValueError("Subpixel(activation=%r): fused activations are not on the path (utils.py:195 passes none)"
                             % (activation,))
        if padding not in ("same", "valid"):
            raise ValueError("Subpixel: padding must be 'same' or 'valid'")
        super().__init__(r * r * filters, kernel_size, strides=strides, padding=padding, use_bias=use_bias,
                         activation=activation, name=name)
        self.r = int(r)
        self.cfg["r"] = int(r)
        self.cfg["out_filters"] = int(filters)

    def compute_output_shape(self, in_shapes):
        # subpixel.py:93-95
        h, w, c = super().compute_output_shape(in_shapes)
        return (self.r * h, self.r * w, c // (self.r * self.r))


class ICNR:
    """ICNR initialiser (subpixel.py:13-39).  `initializer(shape) -> ndarray` draws the sub-kernel."""

    def __init__(self, initializer, scale=1):
        self.scale = scale
        self.initializer = initializer

    def __call__(self, shape, dtype=np.float32, partition_info=None):
        shape = list(shape)
        if self.scale == 1:
            return np.asarray(self.initializer(shape), dtype)
        s = self.scale
        kh, kw, cin, cout = shape
        n = cout // (s * s)
        x = np.asarray(self.initializer([kh, kw, cin, n]), dtype)
        x = x.transpose(2, 0, 1, 3)                                   # [cin, kh, kw, n]
        # tf.image.resize_nearest_neighbor (legacy): src = floor(dst * in/out)
        ys = np.floor(np.arange(kh * s) * (kh / float(kh * s))).astype(np.int64)
        xs = np.floor(np.arange(kw * s) * (kw / float(kw * s))).astype(np.int64)
        x = x[:, ys][:, :, xs]                                        # [cin, kh*s, kw*s, n]
        # tf.space_to_depth(block_size=s): channel = (dy*s + dx)*n + c
        x = x.reshape(cin, kh, s, kw, s, n).transpose(0, 1, 3, 2, 4, 5).reshape(cin, kh, kw, s * s * n)
        return np.ascontiguousarray(x.transpose(1, 2, 0, 3))          # [kh, kw, cin, cout]


def _glorot_normal(shape):
    kh, kw, cin, cout = shape
    return G.glorot_normal(tuple(shape), kh * kw * cin, kh * kw * cout)


def icnr_weights(init=_glorot_normal, scale=2, shape=(3, 3, 32, 4), dtype=np.float32):
    """subpixel.py:9-11 (the reference evaluates the initialiser in a throw-away tf.Session)."""
    return ICNR(init, scale=scale)(shape=shape, dtype=dtype)
