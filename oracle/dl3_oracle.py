"""CPU oracle for the DeepLabV3+ forward/backward path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of what the reference computes when `Deeplabv3()` (deeplabv3p.py:209-466),
`Subpixel` (subpixel.py:41-103), the `SegModel` heads (utils.py:169-214) and the loss
`sparse_crossentropy_ignoring_last_label` (utils.py:127-130) run under Keras 2.2.4 / TF 1.13.

PARITY UNPINNED: the arithmetic of the reference lives in TensorFlow/Keras, which are un-vendored,
un-pinned pip dependencies (README.md:41-44) absent from this container, and the reference ships
no tests or golden vectors.  The TF semantics encoded here (SAME padding, legacy bilinear resize,
FusedBatchNorm, Keras cross-entropy) are restated from their published behaviour and are pinned
only by (a) analytic known-answer tests and (b) an independent torch-CPU restatement
(oracle/torch_ref.py) that must agree — see tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Layout NHWC; weights in Keras layouts; `params` is a dict keyed by Keras weight names
("<layer>/kernel:0", "<layer>/depthwise_kernel:0", "<layer>/bias:0", "<layer>/gamma:0", ...).
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# minimal reverse-mode tape
# --------------------------------------------------------------------------------------


class Tape:
    """Records (output, inputs, backward-closure) triples; `backward` walks them in reverse."""

    def __init__(self):
        self.nodes = []

    def record(self, out, inputs, bwd):
        self.nodes.append((out, inputs, bwd))

    def backward(self, out, gout, wrt=()):
        grads = {id(out): gout}
        keep = {}
        for o, inputs, bwd in reversed(self.nodes):
            g = grads.pop(id(o), None)
            if g is None:
                continue
            gins = bwd(g)
            for t, gi in zip(inputs, gins):
                if gi is None:
                    continue
                if id(t) in grads:
                    grads[id(t)] = grads[id(t)] + gi
                else:
                    grads[id(t)] = gi
                keep[id(t)] = t
        return grads


def _rec(tape, out, inputs, bwd):
    if tape is not None:
        tape.record(out, inputs, bwd)
    return out


# --------------------------------------------------------------------------------------
# padding rules
# --------------------------------------------------------------------------------------


def same_pads(size, k, stride, rate):
    """[TF-semantics] Keras padding='same': out = ceil(in/s); total = max((out-1)s+(k-1)r+1-in, 0);
    begin = total//2 (the odd pixel goes to the END).  Used by deeplabv3p.py:61-62,:98-104,:186-188,:283,:318."""
    out = -(-size // stride)
    keff = (k - 1) * rate + 1
    total = max((out - 1) * stride + keff - size, 0)
    beg = total // 2
    return out, beg, total - beg


def explicit_pads(size, k, stride, rate):
    """deeplabv3p.py:63-69,:105-116: ZeroPadding2D((pad_beg, pad_end)) + 'valid' for stride != 1.
    Keras reads the 2-tuple as symmetric (H, W) pads = ((pad_beg,pad_beg),(pad_end,pad_end)); both
    readings coincide for the only cases that occur (k=3,r=1 -> 1/1; k=1 -> 0/0)."""
    keff = k + (k - 1) * (rate - 1)
    total = keff - 1
    beg = total // 2
    end = total - beg
    out = (size + beg + end - keff) // stride + 1
    return out, beg, end


def _shifted(x, Ho, Wo, stride, oy0, ox0):
    """view v[n,oy,ox,c] = x[n, oy*stride+oy0, ox*stride+ox0, c] with zeros outside; returns (view, slices)"""
    N, H, W, C = x.shape
    # smallest oy with oy*stride+oy0 >= 0
    oy_lo = 0 if oy0 >= 0 else (-oy0 + stride - 1) // stride
    ox_lo = 0 if ox0 >= 0 else (-ox0 + stride - 1) // stride
    oy_hi = min(Ho, (H - 1 - oy0) // stride + 1) if H - 1 - oy0 >= 0 else 0
    ox_hi = min(Wo, (W - 1 - ox0) // stride + 1) if W - 1 - ox0 >= 0 else 0
    if oy_hi <= oy_lo or ox_hi <= ox_lo:
        return None
    ys = slice(oy_lo * stride + oy0, (oy_hi - 1) * stride + oy0 + 1, stride)
    xs = slice(ox_lo * stride + ox0, (ox_hi - 1) * stride + ox0 + 1, stride)
    return (slice(oy_lo, oy_hi), slice(ox_lo, ox_hi)), (ys, xs)


# --------------------------------------------------------------------------------------
# operators (forward + backward)
# --------------------------------------------------------------------------------------


def depthwise3x3(x, w, stride, rate, pad_t, pad_l, Ho, Wo, tape=None):
    """DepthwiseConv2D (deeplabv3p.py:73-74,:186-188), depth multiplier 1, no bias.
    y[n,oy,ox,c] = sum_ij x[n, oy*s-pt+i*r, ox*s-pl+j*r, c] * w[i,j,c]      w: (3,3,C)"""
    N, H, W, C = x.shape
    y = np.zeros((N, Ho, Wo, C), x.dtype)
    taps = []
    for i in range(3):
        for j in range(3):
            sl = _shifted(x, Ho, Wo, stride, i * rate - pad_t, j * rate - pad_l)
            if sl is None:
                continue
            (oys, oxs), (ys, xs) = sl
            y[:, oys, oxs, :] += x[:, ys, xs, :] * w[i, j]
            taps.append((i, j, oys, oxs, ys, xs))

    def bwd(g):
        dx = np.zeros_like(x)
        dw = np.zeros_like(w)
        for i, j, oys, oxs, ys, xs in taps:
            gs = g[:, oys, oxs, :]
            dx[:, ys, xs, :] += gs * w[i, j]
            dw[i, j] = np.einsum("nhwc,nhwc->c", x[:, ys, xs, :], gs)
        return dx, dw

    return _rec(tape, y, (x, w), bwd)


def conv2d(x, w, stride, pad_t, pad_l, Ho, Wo, bias=None, tape=None):
    """Conv2D, cross-correlation, HWIO kernel (deeplabv3p.py:99-116,:283,:318,:377,...).  k = w.shape[0] (1 or 3)."""
    N, H, W, Cin = x.shape
    k = w.shape[0]
    Cout = w.shape[3]
    y = np.zeros((N, Ho, Wo, Cout), x.dtype)
    taps = []
    for i in range(k):
        for j in range(k):
            sl = _shifted(x, Ho, Wo, stride, i - pad_t, j - pad_l)
            if sl is None:
                continue
            (oys, oxs), (ys, xs) = sl
            y[:, oys, oxs, :] += x[:, ys, xs, :] @ w[i, j]
            taps.append((i, j, oys, oxs, ys, xs))
    if bias is not None:
        y += bias

    def bwd(g):
        dx = np.zeros_like(x)
        dw = np.zeros_like(w)
        for i, j, oys, oxs, ys, xs in taps:
            gs = g[:, oys, oxs, :]
            dx[:, ys, xs, :] += gs @ w[i, j].T
            xs_ = x[:, ys, xs, :]
            dw[i, j] = xs_.reshape(-1, Cin).T @ gs.reshape(-1, Cout)
        if bias is not None:
            return dx, dw, g.sum(axis=(0, 1, 2))
        return dx, dw

    ins = (x, w) if bias is None else (x, w, bias)
    return _rec(tape, y, ins, bwd)


# [TF-semantics] tensorflow/python/ops/nn_impl.py fused_batch_norm (TF 1.13): `min_epsilon = 1.001e-5;
# epsilon = epsilon if epsilon > min_epsilon else min_epsilon` (a cuDNN requirement the Python wrapper applies on every
# device).  Keras 2.2.4 reaches it for every 4-D NHWC BatchNormalization, in both phases: tensorflow_backend.py
# normalize_batch_in_training -> _fused_normalize_batch_in_training (training) and batch_normalization ->
# tf.nn.fused_batch_norm(is_training=False) (inference).  It only bites the 1e-5 layers of the ASPP / decoder
# (deeplabv3p.py:379,386,393-399,408,422-423,427-429): 1e-5 -> 1.001e-5.  The moving-variance factor of
# BatchNormalization.call uses the layer's own `self.epsilon`, un-floored.
TF_FUSED_BN_MIN_EPSILON = 1.001e-5


def fused_bn_epsilon(eps):
    """the epsilon tf.nn.fused_batch_norm normalises with (see TF_FUSED_BN_MIN_EPSILON)"""
    return eps if eps > TF_FUSED_BN_MIN_EPSILON else TF_FUSED_BN_MIN_EPSILON


def batchnorm(x, gamma, beta, mmean, mvar, eps, training, momentum=0.99, tape=None, stats_out=None):
    """BatchNormalization over the last axis.
    inference: gamma*(x-mm)/sqrt(mv+eps_f)+beta, eps_f = fused_bn_epsilon(eps).
    training [TF-semantics FusedBatchNorm]: biased batch variance over (N,H,W) for the
    normalisation; moving = m*moving + (1-m)*batch.  The variance that enters the moving average, in the reference's
    environment (Keras 2.2.4 on TF 1.13): tf.nn.fused_batch_norm returns the Bessel-corrected batch
    variance (n/(n-1)) and keras/layers/normalization.py BatchNormalization.call then applies
    `variance *= sample_size / (sample_size - (1.0 + self.epsilon))` on top of it — both factors are restated here
    (from memory of those two sources; neither package is installable in this container).
    `stats_out` (dict) receives the updated moving statistics (side output, no gradient)."""
    axes = tuple(range(x.ndim - 1))
    if training:
        M = x.size // x.shape[-1]
        mean = x.mean(axis=axes, dtype=np.float64)
        var = ((x.astype(np.float64) - mean) ** 2).mean(axis=axes)
        invstd = 1.0 / np.sqrt(var + fused_bn_epsilon(eps))
        xhat = ((x - mean) * invstd).astype(x.dtype)
        y = (xhat * gamma + beta).astype(x.dtype)
        if stats_out is not None:
            unb = var * M / (M - 1) if M > 1 else var
            if M > 1.0 + eps:
                unb = unb * (M / (M - (1.0 + eps)))
            stats_out["mean"] = (momentum * mmean + (1 - momentum) * mean).astype(x.dtype)
            stats_out["var"] = (momentum * mvar + (1 - momentum) * unb).astype(x.dtype)
            stats_out["batch_mean"] = mean
            stats_out["batch_var"] = var

        def bwd(g):
            dbeta = g.sum(axis=axes, dtype=np.float64)
            dgamma = (g * xhat).sum(axis=axes, dtype=np.float64)
            dx = (gamma * invstd) * (g - dbeta / M - xhat * (dgamma / M))
            return dx.astype(x.dtype), dgamma.astype(x.dtype), dbeta.astype(x.dtype)

        return _rec(tape, y, (x, gamma, beta), bwd)
    invstd = 1.0 / np.sqrt(mvar.astype(np.float64) + fused_bn_epsilon(eps))
    xhat = ((x - mmean) * invstd).astype(x.dtype)
    y = (xhat * gamma + beta).astype(x.dtype)

    def bwd(g):
        dbeta = g.sum(axis=axes, dtype=np.float64)
        dgamma = (g * xhat).sum(axis=axes, dtype=np.float64)
        return (g * (gamma * invstd)).astype(x.dtype), dgamma.astype(x.dtype), dbeta.astype(x.dtype)

    return _rec(tape, y, (x, gamma, beta), bwd)


def relu(x, tape=None):
    y = np.maximum(x, 0)
    return _rec(tape, y, (x,), lambda g: (g * (x > 0),))


def relu6(x, tape=None):
    """relu(x, max_value=6.) (deeplabv3p.py:181,:192,:325)"""
    y = np.minimum(np.maximum(x, 0), 6)
    return _rec(tape, y, (x,), lambda g: (g * ((x > 0) & (x < 6)),))


def add(a, b, tape=None):
    return _rec(tape, a + b, (a, b), lambda g: (g, g))


def concat(xs, tape=None):
    sizes = np.cumsum([0] + [t.shape[-1] for t in xs])
    y = np.concatenate(xs, axis=-1)
    return _rec(tape, y, tuple(xs), lambda g: tuple(g[..., sizes[i]:sizes[i + 1]] for i in range(len(xs))))


def global_avg_pool(x, tape=None):
    """AveragePooling2D with pool == feature map (deeplabv3p.py:375): [N,H,W,C] -> [N,1,1,C]"""
    N, H, W, C = x.shape
    y = x.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(x.dtype)
    return _rec(tape, y, (x,), lambda g: (np.broadcast_to(g / (H * W), x.shape).astype(x.dtype),))


def _tf1_lerp(out_size, in_size):
    """[TF-semantics] tf.image.resize_bilinear, TF 1.x legacy (align_corners=False, no half-pixel
    centres): src = dst*(in/out) in fp32; lower=floor(src); upper=min(lower+1,in-1); lerp=src-lower."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.arange(out_size, dtype=np.float32) * scale
    lo = np.floor(src).astype(np.int64)
    lo = np.minimum(lo, in_size - 1)
    hi = np.minimum(lo + 1, in_size - 1)
    w = (src - lo.astype(np.float32)).astype(np.float32)
    return lo, hi, w


def resize_bilinear_tf1(x, Ho, Wo, tape=None):
    """Lambda(K.tf.image.resize_bilinear) at deeplabv3p.py:382,:418,:439 and utils.py:190."""
    N, Hi, Wi, C = x.shape
    ylo, yhi, wy = _tf1_lerp(Ho, Hi)
    xlo, xhi, wx = _tf1_lerp(Wo, Wi)
    wx_ = wx[None, None, :, None].astype(x.dtype)
    wy_ = wy[None, :, None, None].astype(x.dtype)
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * wx_
    bot = bl + (br - bl) * wx_
    y = top + (bot - top) * wy_

    def bwd(g):
        dx = np.zeros_like(x)
        gtop = g * (1 - wy_)
        gbot = g * wy_
        for (yi, gg) in ((ylo, gtop), (yhi, gbot)):
            for (xi, ww) in ((xlo, 1 - wx_), (xhi, wx_)):
                np.add.at(dx, (slice(None), yi[:, None], xi[None, :]), gg * ww)
        return (dx,)

    return _rec(tape, y.astype(x.dtype), (x,), bwd)


def dropout(x, mask, rate, tape=None):
    """Dropout(0.1) (deeplabv3p.py:410) in training mode with a SUPPLIED keep mask (TF's RNG stream
    is not reproducible): y = x*mask/(1-rate)."""
    s = x.dtype.type(1.0 / (1.0 - rate))
    return _rec(tape, x * mask * s, (x,), lambda g: (g * mask * s,))


def phase_shift(I, r, tape=None):
    """Subpixel._phase_shift (subpixel.py:77-88): out[n, ia*r+q, ib*r+p, ch] = I[n, ia, ib, ch*r*r + p*r + q]."""
    N, a, b, c = I.shape
    co = c // (r * r)
    X = I.reshape(N, a, b, co, r, r)          # [n, ia, ib, ch, p, q]
    X = X.transpose(0, 1, 5, 2, 4, 3)        # [n, ia, q, ib, p, ch]
    y = np.ascontiguousarray(X.reshape(N, a * r, b * r, co))

    def bwd(g):
        G = g.reshape(N, a, r, b, r, co).transpose(0, 1, 3, 5, 4, 2)  # [n, ia, ib, ch, p, q]
        return (np.ascontiguousarray(G.reshape(N, a, b, c)),)

    return _rec(tape, y, (I,), bwd)


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def loss_sparse_xent_ignoring_last_label(logits, labels, weights):
    """utils.py:127-130 + Keras temporal sample weights (notebook cell 2, sample_weight_mode='temporal').
    logits [B, HW, C] (pre-softmax), labels [B, HW] (float, void == C), weights [B, HW].
    [TF-semantics] Keras categorical_crossentropy on probabilities: p /= sum(p); clip(p,1e-7,1-1e-7);
    l = -sum(y*log p); weighted: score = l*w; score /= mean(w != 0); loss = mean(score)
      => loss = sum(l*w)/count(w != 0).
    Returns (loss, dlogits) with dlogits = (p - onehot)*w/nnz where the true-class probability lies inside the clip
    interval [1e-7, 1-1e-7] and ZERO where it does not: [TF-semantics] the gradient of tf.clip_by_value is zero outside
    its bounds, so such a pixel's loss is a constant (round 4; before, the clip was ignored in the gradient — identical
    unless a probability has left the interval, i.e. a logit gap above ~16).  A void row (label == C) has an all-zero one-hot row (utils.py:129 drops the last
    column), so it contributes neither loss nor gradient — whatever its sample weight; a non-zero weight there still
    counts in nnz, exactly as Keras' mean(w != 0) does."""
    C = logits.shape[-1]
    p = softmax(logits)
    t = labels.astype(np.int64)
    onehot = np.zeros(p.shape, p.dtype)
    valid = (t >= 0) & (t < C)
    bi, pi = np.nonzero(valid)
    onehot[bi, pi, t[bi, pi]] = 1
    q = p / p.sum(axis=-1, keepdims=True)
    qt = (onehot * q).sum(axis=-1)
    inside = (qt >= 1e-7) & (qt <= 1 - 1e-7)  # clip_by_value passes the gradient here (bounds included) and only here
    q = np.clip(q, 1e-7, 1 - 1e-7)
    l = -(onehot * np.log(q)).sum(axis=-1)
    nnz = max(float((weights != 0).sum()), 1.0)
    loss = float((l * weights).sum(dtype=np.float64) / nnz)
    dlogits = (p - onehot) * (weights * (valid & inside) / p.dtype.type(nnz))[..., None]
    return loss, dlogits.astype(logits.dtype), p


# --------------------------------------------------------------------------------------
# initialisers
# --------------------------------------------------------------------------------------


def _make_divisible(v, divisor, min_value=None):
    """deeplabv3p.py:157-164"""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def icnr_from_subkernel(X, scale):
    """ICNR.__call__ (subpixel.py:27-39) applied to an already drawn sub-kernel X [kh,kw,Cin,n]:
    transpose[2,0,1,3] -> nearest-neighbour resize x scale -> space_to_depth(scale) -> transpose[1,2,0,3].
    [TF-semantics] resize_nearest_neighbor (legacy): src = floor(dst*in/out); space_to_depth output
    channel = (dy*scale + dx)*n + c.  For 1x1 kernels this gives W[0,0,i,k] = X[0,0,i,k % n]."""
    kh, kw, cin, n = X.shape
    x = X.transpose(2, 0, 1, 3)  # [cin, kh, kw, n]
    ys = np.floor(np.arange(kh * scale) * (kh / (kh * scale))).astype(int)
    xs = np.floor(np.arange(kw * scale) * (kw / (kw * scale))).astype(int)
    up = x[:, ys][:, :, xs]  # [cin, kh*s, kw*s, n]
    s2d = up.reshape(cin, kh, scale, kw, scale, n).transpose(0, 1, 3, 2, 4, 5).reshape(cin, kh, kw, scale * scale * n)
    return np.ascontiguousarray(s2d.transpose(1, 2, 0, 3))


# --------------------------------------------------------------------------------------
# model restatement (functional; follows deeplabv3p.py top to bottom)
# --------------------------------------------------------------------------------------

MNV2_BLOCKS = [
    # (block_id, filters, stride, expansion, skip, rate)      deeplabv3p.py:327-367
    (0, 16, 1, 1, False, 1), (1, 24, 2, 6, False, 1), (2, 24, 1, 6, True, 1), (3, 32, 2, 6, False, 1),
    (4, 32, 1, 6, True, 1), (5, 32, 1, 6, True, 1), (6, 64, 1, 6, False, 1), (7, 64, 1, 6, True, 2),
    (8, 64, 1, 6, True, 2), (9, 64, 1, 6, True, 2), (10, 96, 1, 6, False, 2), (11, 96, 1, 6, True, 2),
    (12, 96, 1, 6, True, 2), (13, 160, 1, 6, False, 2), (14, 160, 1, 6, True, 4), (15, 160, 1, 6, True, 4),
    (16, 320, 1, 6, False, 4),
]


# OS -> (entry_block3_stride, middle_block_rate, exit_block_rates, atrous_rates)      deeplabv3p.py:272-282
XCEPTION_OS = {8: (1, 2, (2, 4), (12, 24, 36)), 16: (2, 1, (1, 2), (6, 12, 18))}


class _Net:
    """Carries params/tape/mode through the functional model and collects BN side outputs."""

    def __init__(self, params, training, tape, dropout_mask=None, bn_frozen=False):
        self.p = params
        self.training = training
        self.tape = tape
        self.dropout_mask = dropout_mask
        self.bn_frozen = bn_frozen
        self.new_stats = {}
        self.used = []
        self.views = {}

    def W(self, name):
        self.used.append(name)
        return self.p[name]

    def conv(self, x, name, k=1, stride=1, rate=1, same=True, bias=False):
        w = self.W(name + "/kernel:0")
        N, H, Wd, _ = x.shape
        if k == 1 and stride > 1:
            Ho, pt, _ = explicit_pads(H, 1, stride, 1)
            Wo, pl, _ = explicit_pads(Wd, 1, stride, 1)
        elif same:
            Ho, pt, _ = same_pads(H, k, stride, rate)
            Wo, pl, _ = same_pads(Wd, k, stride, rate)
        else:
            Ho, pt, _ = explicit_pads(H, k, stride, rate)
            Wo, pl, _ = explicit_pads(Wd, k, stride, rate)
        assert rate == 1
        b = self.W(name + "/bias:0") if bias else None
        return conv2d(x, w, stride, pt, pl, Ho, Wo, bias=b, tape=self.tape)

    def dw(self, x, name, stride=1, rate=1, same=True):
        w = self.views.setdefault(name, self.W(name + "/depthwise_kernel:0")[..., 0])
        N, H, Wd, _ = x.shape
        if same:
            Ho, pt, _ = same_pads(H, 3, stride, rate)
            Wo, pl, _ = same_pads(Wd, 3, stride, rate)
        else:
            Ho, pt, _ = explicit_pads(H, 3, stride, rate)
            Wo, pl, _ = explicit_pads(Wd, 3, stride, rate)
        return depthwise3x3(x, w, stride, rate, pt, pl, Ho, Wo, tape=self.tape)

    def bn(self, x, name, eps=1e-3, momentum=0.99):
        st = {}
        y = batchnorm(x, self.W(name + "/gamma:0"), self.W(name + "/beta:0"), self.W(name + "/moving_mean:0"),
                      self.W(name + "/moving_variance:0"), eps, self.training and not self.bn_frozen,
                      momentum=momentum, tape=self.tape, stats_out=st)
        if st:
            self.new_stats[name] = st
        return y


def _sepconv_bn(net, x, filters, prefix, stride=1, rate=1, depth_activation=False, eps=1e-3):
    """SepConv_BN (deeplabv3p.py:47-84)"""
    if not depth_activation:
        x = relu(x, net.tape)
    x = net.dw(x, prefix + "_depthwise", stride=stride, rate=rate, same=(stride == 1))
    x = net.bn(x, prefix + "_depthwise_BN", eps=eps)
    if depth_activation:
        x = relu(x, net.tape)
    x = net.conv(x, prefix + "_pointwise")
    x = net.bn(x, prefix + "_pointwise_BN", eps=eps)
    if depth_activation:
        x = relu(x, net.tape)
    return x


def _xception_block(net, inputs, depth_list, prefix, skip_type, stride, rate=1, depth_activation=False,
                    return_skip=False):
    """_xception_block (deeplabv3p.py:119-155); `layers.add` == Add."""
    residual = inputs
    skip = None
    for i in range(3):
        residual = _sepconv_bn(net, residual, depth_list[i], prefix + "_separable_conv%d" % (i + 1),
                               stride=stride if i == 2 else 1, rate=rate, depth_activation=depth_activation)
        if i == 1:
            skip = residual
    if skip_type == "conv":
        shortcut = net.conv(inputs, prefix + "_shortcut", k=1, stride=stride)
        shortcut = net.bn(shortcut, prefix + "_shortcut_BN")
        out = add(residual, shortcut, net.tape)
    elif skip_type == "sum":
        out = add(residual, inputs, net.tape)
    else:
        out = residual
    return (out, skip) if return_skip else out


def deeplab_features(net, x, backbone, input_shape, OS, alpha, classes_head=True):
    """Everything of Deeplabv3() up to and including Dropout (deeplabv3p.py:270-429) —
    i.e. the output of model.layers[-5] that utils.py:181 cuts at."""
    H, W = input_shape[0], input_shape[1]
    tape = net.tape
    # Lambda(x/127.5 - 1)  deeplabv3p.py:270
    xin = x
    x = (xin / xin.dtype.type(127.5) - xin.dtype.type(1)).astype(xin.dtype)
    _rec(tape, x, (xin,), lambda g: (g / xin.dtype.type(127.5),))
    skip1 = None
    if backbone == "xception":
        entry3_stride, middle_rate, exit_rates, atrous = XCEPTION_OS[8 if OS == 8 else 16]
        x = net.conv(x, "entry_flow_conv1_1", k=3, stride=2)
        x = relu(net.bn(x, "entry_flow_conv1_1_BN"), tape)
        x = net.conv(x, "entry_flow_conv1_2", k=3, stride=1)
        x = relu(net.bn(x, "entry_flow_conv1_2_BN"), tape)
        x = _xception_block(net, x, [128, 128, 128], "entry_flow_block1", "conv", 2)
        x, skip1 = _xception_block(net, x, [256, 256, 256], "entry_flow_block2", "conv", 2, return_skip=True)
        x = _xception_block(net, x, [728, 728, 728], "entry_flow_block3", "conv", entry3_stride)
        for i in range(16):
            x = _xception_block(net, x, [728, 728, 728], "middle_flow_unit_%d" % (i + 1), "sum", 1, rate=middle_rate)
        x = _xception_block(net, x, [728, 1024, 1024], "exit_flow_block1", "conv", 1, rate=exit_rates[0])
        x = _xception_block(net, x, [1536, 1536, 2048], "exit_flow_block2", "none", 1, rate=exit_rates[1],
                            depth_activation=True)
    else:
        OS = 8  # deeplabv3p.py:316 — MobileNetV2 silently runs at output stride 8
        x = net.conv(x, "Conv", k=3, stride=2)
        x = relu6(net.bn(x, "Conv_BN", momentum=0.999), tape)
        for bid, filters, stride, expansion, skip, rate in MNV2_BLOCKS:
            inp = x
            prefix = "expanded_conv_%d_" % bid if bid else "expanded_conv_"
            if bid:
                x = net.conv(x, prefix + "expand")
                x = relu6(net.bn(x, prefix + "expand_BN", momentum=0.999), tape)
            x = net.dw(x, prefix + "depthwise", stride=stride, rate=rate)
            x = relu6(net.bn(x, prefix + "depthwise_BN", momentum=0.999), tape)
            x = net.conv(x, prefix + "project")
            x = net.bn(x, prefix + "project_BN", momentum=0.999)
            if skip:
                x = add(inp, x, tape)
    fh, fw = int(np.ceil(H / OS)), int(np.ceil(W / OS))
    assert x.shape[1] == fh and x.shape[2] == fw, (x.shape, fh, fw)
    # ASPP  deeplabv3p.py:375-410
    b4 = global_avg_pool(x, tape)
    b4 = net.conv(b4, "image_pooling")
    b4 = relu(net.bn(b4, "image_pooling_BN", eps=1e-5), tape)
    b4 = resize_bilinear_tf1(b4, fh, fw, tape)
    b0 = net.conv(x, "aspp0")
    b0 = relu(net.bn(b0, "aspp0_BN", eps=1e-5), tape)
    if backbone == "xception":
        bs = [_sepconv_bn(net, x, 256, "aspp%d" % (i + 1), rate=atrous[i], depth_activation=True, eps=1e-5)
              for i in range(3)]
        x = concat([b4, b0] + bs, tape)
    else:
        x = concat([b4, b0], tape)
    x = net.conv(x, "concat_projection")
    x = relu(net.bn(x, "concat_projection_BN", eps=1e-5), tape)
    if net.training and net.dropout_mask is not None:
        x = dropout(x, net.dropout_mask, 0.1, tape)
    if backbone == "xception":
        x = resize_bilinear_tf1(x, int(np.ceil(H / 4)), int(np.ceil(W / 4)), tape)
        d = net.conv(skip1, "feature_projection0")
        d = relu(net.bn(d, "feature_projection0_BN", eps=1e-5), tape)
        x = concat([x, d], tape)
        x = _sepconv_bn(net, x, 256, "decoder_conv0", depth_activation=True, eps=1e-5)
        x = _sepconv_bn(net, x, 256, "decoder_conv1", depth_activation=True, eps=1e-5)
    return x


def forward(params, x, backbone="mobilenetv2", input_shape=(512, 512, 3), classes=21, OS=16, alpha=1.0,
            head="deeplab", training=False, tape=None, dropout_mask=None, bn_frozen=False, subpixel_name="subpixel_1"):
    """Pre-softmax logits [B, H, W, classes] of
       head='deeplab' : Deeplabv3() itself (deeplabv3p.py:432-439, layer logits_semantic/custom_logits_semantic)
       head='original': SegModel 'original' head (utils.py:189-190, conv_upsample + bilinear)
       head='subpixel': SegModel 'subpixel' head (utils.py:195, Subpixel(n,1,scale))
    Returns (logits, net)."""
    assert alpha == 1.0 or backbone == "mobilenetv2"
    net = _Net(params, training, tape, dropout_mask, bn_frozen)
    H, W = input_shape[0], input_shape[1]
    f = deeplab_features(net, x, backbone, input_shape, OS, alpha)
    if head == "subpixel":
        scale = 4 if backbone == "xception" else 8
        y = net.conv(f, subpixel_name, bias=True)
        y = phase_shift(y, scale, tape)
    else:
        name = "conv_upsample" if head == "original" else ("logits_semantic" if classes == 21 else "custom_logits_semantic")
        y = net.conv(f, name, bias=True)
        y = resize_bilinear_tf1(y, H, W, tape)
    return y, net


def train_grads(params, x, labels, weights, **kw):
    """One forward + backward: returns (loss, {weight name: gradient}, logits, net)."""
    tape = Tape()
    logits, net = forward(params, x, training=True, tape=tape, **kw)
    B, H, W, C = logits.shape
    loss, dl, _ = loss_sparse_xent_ignoring_last_label(logits.reshape(B, H * W, C), labels.reshape(B, H * W),
                                                       weights.reshape(B, H * W))
    grads = tape.backward(logits, dl.reshape(logits.shape))
    out = {}
    for name in set(net.used):
        arr = params[name]
        g = grads.get(id(arr))
        if name.endswith("/depthwise_kernel:0"):
            g = grads.get(id(net.views[name[: -len("/depthwise_kernel:0")]]))
            g = None if g is None else g[..., None]
        out[name] = g
    return loss, out, logits, net


# --------------------------------------------------------------------------------------
# the optimizer of the reference's training cell: Adam(lr=7e-4, epsilon=1e-8, decay=1e-6)
# (segmentation.ipynb cell 2, json line 107) as Keras 2.2.4 keras/optimizers.py Adam.get_updates states it
# --------------------------------------------------------------------------------------
ADAM_DEFAULTS = dict(lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6)


def adam_update(p, g, m, v, iterations, lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6):
    """One Keras 2.2.4 Adam update of one weight [TF-semantics, Adam.get_updates]:
        lr   = lr * 1 / (1 + decay * iterations)          (only when decay > 0; `iterations` BEFORE its increment)
        t    = iterations + 1
        lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)
        m_t  = beta_1 * m + (1 - beta_1) * g ;  v_t = beta_2 * v + (1 - beta_2) * g^2
        p_t  = p - lr_t * m_t / (sqrt(v_t) + epsilon)      (epsilon OUTSIDE the root, no m_hat / v_hat division)
    Returns (p_t, m_t, v_t)."""
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * iterations))
    t = iterations + 1
    lr_t = lr * (np.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t))
    m_t = beta_1 * m + (1.0 - beta_1) * g
    v_t = beta_2 * v + (1.0 - beta_2) * np.square(g)
    return p - lr_t * m_t / (np.sqrt(v_t) + epsilon), m_t, v_t


def train_steps(params, batches, opt=None, frozen=(), **kw):
    """len(batches) x Model.train_on_batch (notebook json 107,160: fit_generator -> train_on_batch): forward in the
    training phase, loss, gradients, one Adam update of every trainable weight, the BatchNorm moving-statistics update.
    batches: [(x, labels, weights), ...]; frozen: weight names that do not train (layer.trainable = False, notebook json
    147-155); kw as for train_grads (bn_frozen=True: BatchNorm normalises with the moving statistics, which then stay).
    Returns (losses, params after the last step, {name: (m, v)})."""
    o = dict(ADAM_DEFAULTS)
    o.update(opt or {})
    p = dict(params)
    state = {}
    losses = []
    for it, (x, y, w) in enumerate(batches):
        loss, grads, _, net = train_grads(p, x, y, w, **kw)
        losses.append(float(loss))
        new = dict(p)
        for name, g in grads.items():
            if g is None or name in frozen or name.split("/")[-1].startswith("moving"):
                continue
            m, v = state.get(name, (np.zeros_like(p[name]), np.zeros_like(p[name])))
            new[name], m, v = adam_update(p[name], g.reshape(p[name].shape), m, v, it, **o)
            state[name] = (m, v)
        for bn, st in net.new_stats.items():
            if (bn + "/gamma:0") in frozen:   # Keras 2.2.x collects no updates from a non-trainable layer
                continue
            new[bn + "/moving_mean:0"] = st["mean"]
            new[bn + "/moving_variance:0"] = st["var"]
        p = new
    return losses, p, state


# --------------------------------------------------------------------------------------
# parameter construction (shapes follow the Keras layers the reference instantiates)
# --------------------------------------------------------------------------------------


def param_shapes(backbone="mobilenetv2", classes=21, OS=16, alpha=1.0, head="deeplab", subpixel_name="subpixel_1"):
    """Ordered {weight name: shape} for every weight the model owns."""
    shapes = {}

    def conv(name, k, cin, cout, bias=False):
        shapes[name + "/kernel:0"] = (k, k, cin, cout)
        if bias:
            shapes[name + "/bias:0"] = (cout,)

    def dw(name, c):
        shapes[name + "/depthwise_kernel:0"] = (3, 3, c, 1)

    def bn(name, c):
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            shapes["%s/%s:0" % (name, v)] = (c,)

    def sep(prefix, cin, cout):
        dw(prefix + "_depthwise", cin)
        bn(prefix + "_depthwise_BN", cin)
        conv(prefix + "_pointwise", 1, cin, cout)
        bn(prefix + "_pointwise_BN", cout)

    if backbone == "xception":
        conv("entry_flow_conv1_1", 3, 3, 32); bn("entry_flow_conv1_1_BN", 32)
        conv("entry_flow_conv1_2", 3, 32, 64); bn("entry_flow_conv1_2_BN", 64)
        c = 64

        def block(prefix, depths, skip):
            nonlocal c
            cin = c
            for i, d in enumerate(depths):
                sep(prefix + "_separable_conv%d" % (i + 1), c, d)
                c = d
            if skip == "conv":
                conv(prefix + "_shortcut", 1, cin, depths[-1]); bn(prefix + "_shortcut_BN", depths[-1])

        block("entry_flow_block1", [128] * 3, "conv")
        block("entry_flow_block2", [256] * 3, "conv")
        block("entry_flow_block3", [728] * 3, "conv")
        for i in range(16):
            block("middle_flow_unit_%d" % (i + 1), [728] * 3, "sum")
        block("exit_flow_block1", [728, 1024, 1024], "conv")
        block("exit_flow_block2", [1536, 1536, 2048], "none")
    else:
        first = _make_divisible(32 * alpha, 8)
        conv("Conv", 3, 3, first); bn("Conv_BN", first)
        c = first
        for bid, filters, stride, expansion, skip, rate in MNV2_BLOCKS:
            prefix = "expanded_conv_%d_" % bid if bid else "expanded_conv_"
            pf = _make_divisible(int(filters * alpha), 8)
            e = c
            if bid:
                e = expansion * c
                conv(prefix + "expand", 1, c, e); bn(prefix + "expand_BN", e)
            dw(prefix + "depthwise", e); bn(prefix + "depthwise_BN", e)
            conv(prefix + "project", 1, e, pf); bn(prefix + "project_BN", pf)
            c = pf
    conv("image_pooling", 1, c, 256); bn("image_pooling_BN", 256)
    conv("aspp0", 1, c, 256); bn("aspp0_BN", 256)
    ncat = 512
    if backbone == "xception":
        for i in range(3):
            sep("aspp%d" % (i + 1), c, 256)
        ncat = 1280
    conv("concat_projection", 1, ncat, 256); bn("concat_projection_BN", 256)
    if backbone == "xception":
        conv("feature_projection0", 1, 256, 48); bn("feature_projection0_BN", 48)
        sep("decoder_conv0", 304, 256)
        sep("decoder_conv1", 256, 256)
    if head == "subpixel":
        scale = 4 if backbone == "xception" else 8
        conv(subpixel_name, 1, 256, classes * scale * scale, bias=True)
    elif head == "original":
        conv("conv_upsample", 1, 256, classes, bias=True)
    else:
        conv("logits_semantic" if classes == 21 else "custom_logits_semantic", 1, 256, classes, bias=True)
    return shapes


def init_params(shapes, seed=1, dtype=np.float32, bn_random=True):
    """Seeded synthetic weights: Glorot-uniform kernels (Keras fans), zero bias,
    BN gamma~U(.5,1.5), beta~N(0,.1), moving stats mean 0 / var 1 (calibrate separately)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in shapes.items():
        if name.endswith("depthwise_kernel:0"):
            fan_in, fan_out = 9 * shp[2], 9
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            p[name] = rng.uniform(-lim, lim, shp).astype(dtype)
        elif name.endswith("kernel:0"):
            kh, kw, cin, cout = shp
            lim = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            p[name] = rng.uniform(-lim, lim, shp).astype(dtype)
        elif name.endswith("gamma:0"):
            p[name] = (rng.uniform(0.5, 1.5, shp) if bn_random else np.ones(shp)).astype(dtype)
        elif name.endswith("beta:0"):
            p[name] = (rng.normal(0, 0.1, shp) if bn_random else np.zeros(shp)).astype(dtype)
        elif name.endswith("moving_variance:0"):
            p[name] = np.ones(shp, dtype)
        else:  # bias, moving_mean
            p[name] = np.zeros(shp, dtype)
    return p


def calibrate_bn(params, x, **kw):
    """Replace the moving statistics by the batch statistics of one training-mode forward on `x`
    (SURVEY App. C 'practical fixture advice'): keeps activations O(1) through 50-140 layers."""
    _, net = forward(params, x, training=True, **kw)
    for name, st in net.new_stats.items():
        params[name + "/moving_mean:0"] = st["batch_mean"].astype(params[name + "/moving_mean:0"].dtype)
        params[name + "/moving_variance:0"] = st["batch_var"].astype(params[name + "/moving_variance:0"].dtype)
    return params


def jaccard(labels, probs):
    """utils.py:139-157 restated on host (stays on host by decree of the north star):
    per class, IoU per image averaged over the images that contain the class; NaN classes dropped."""
    C = probs.shape[-1]
    pred = probs.argmax(-1)
    ious = []
    for i in range(C):
        t = labels == i
        p = pred == i
        inter = (t & p).sum(axis=1)
        union = (t | p).sum(axis=1)
        legal = t.sum(axis=1) > 0
        if legal.any():
            ious.append(float(np.mean(inter[legal] / union[legal])))
    return float(np.mean(ious)) if ious else float("nan")


# ---------------------------------------------------------------------------------------
# either side of the network
# ---------------------------------------------------------------------------------------
def balanced_class_weights(y):
    """sklearn.utils.class_weight.compute_class_weight('balanced', np.unique(y), y) restated (the reference calls it at
    utils.py:393-395; scikit-learn is un-pinned there): n_samples / (n_classes_present * bincount), float64."""
    y = np.asarray(y)
    classes, counts = np.unique(y, return_counts=True)
    return classes, len(y) / (len(classes) * counts.astype(np.float64))


def prepare_targets(labels, n_classes):
    """The label half of SegmentationGenerator.__getitem__ (utils.py:371-400) for a batch of raw label maps
    labels[B][HW] (any integer dtype): returns Y [B,HW,1] float32, SW [B,HW] float32 and the per-image histogram
    [B, n_classes+1] int32.  (The relabelling of interpolation artefacts at utils.py:372-373 belongs to the
    augmentation, which is out of scope.)"""
    labels = np.asarray(labels)
    B, HW = labels.shape
    Y = np.zeros((B, HW, 1), np.float32)
    SW = np.zeros((B, HW), np.float32)
    hist = np.zeros((B, n_classes + 1), np.int32)
    for n in range(B):
        y = labels[n].astype(np.int32).copy()
        y[(y > n_classes - 1) | (y < 0)] = n_classes                      # utils.py:377
        Y[n] = y[:, None]                                                 # utils.py:379
        hist[n] = np.bincount(y, minlength=n_classes + 1)
        filt_y = y[y != n_classes]                                        # utils.py:391
        if len(filt_y):
            classes, w = balanced_class_weights(filt_y)                   # utils.py:392-395
            for c, wc in zip(classes, w):
                np.putmask(SW[n], y == c, wc)                             # utils.py:397-398 (float64 -> float32)
        np.putmask(SW[n], y == n_classes, 0)                              # utils.py:400
    return Y, SW, hist


def seg_counts(pred, y_true, n_classes):
    """Per-image, per-class pixel counts of utils.py:143-148: [B,3,C] = #(true==c), #(pred==c), #(true==c & pred==c)."""
    pred = np.asarray(pred).reshape(len(pred), -1)
    t = np.asarray(y_true).reshape(len(pred), -1).astype(np.int64)
    out = np.zeros((len(pred), 3, n_classes), np.int32)
    for c in range(n_classes):
        tl, pl = t == c, pred == c
        out[:, 0, c] = tl.sum(1)
        out[:, 1, c] = pl.sum(1)
        out[:, 2, c] = (tl & pl).sum(1)
    return out
