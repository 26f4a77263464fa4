"""Second, independent CPU restatement (torch autograd) — TEST INFRASTRUCTURE ONLY.

Written against the same reference lines as oracle/dl3_oracle.py but with different machinery
(NCHW torch.nn.functional convolutions with explicit F.pad, F.batch_norm, autograd for every
gradient), so that an error in the hand-written numpy forward/backward formulas shows up as a
disagreement (tests/test_oracle.py).  It shares NO code with dl3_oracle: the TF padding rule, the
legacy bilinear resize (here: dense interpolation matrices built pixel by pixel) and the MobileNetV2
block table are restated a second time below, so a slip in one restatement cannot hide in the other.
It also serves as the labelled "framework CPU path" proxy (torch/oneDNN on the host cores) in
bench.py's cpu_baseline leg, because the reference's own Keras/TensorFlow CPU path cannot be
installed here.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F



def _fused_eps(eps):
    """[TF-semantics] tf.nn.fused_batch_norm (nn_impl.py, TF 1.13) never normalises with an epsilon below 1.001e-5; Keras
    2.2.4 runs every 4-D BatchNormalization through it in both phases (own restatement: this file shares no code with
    dl3_oracle.py)"""
    return max(float(eps), 1.001e-5)


def _same(size, k, s, r):
    """[TF-semantics] padding='SAME' (tensorflow/core/framework/common_shape_fns.cc GetWindowedOutputSize):
    the output has ceil(size/s) positions, the input is padded just enough for the last window to fit and the
    extra pixel of an odd total goes AFTER the data."""
    n_out = math.ceil(size / s)
    span = (n_out - 1) * s + (k - 1) * r + 1   # extent covered by all windows (dilated kernel extent k + (k-1)(r-1))
    need = span - size
    if need < 0:
        need = 0
    before = need >> 1
    return before, need - before


def _explicit(size, k, s, r):
    """deeplabv3p.py:63-69 / :105-116 (stride != 1): kernel_size_effective = k + (k-1)(rate-1); pad_total =
    effective - 1; pad_beg = pad_total // 2; pad_end = pad_total - pad_beg; ZeroPadding2D then a VALID conv."""
    eff = k + (k - 1) * (r - 1)
    tot = eff - 1
    return tot // 2, tot - tot // 2


def _resize_matrix(n_out, n_in, dtype):
    """[TF-semantics] tf.image.resize_bilinear of TF 1.x (align_corners=False, no half-pixel centres,
    tensorflow/core/kernels/resize_bilinear_op.cc compute_interpolation_weights): for output index i the source
    coordinate is i * (n_in / n_out) evaluated in float32; it blends floor(coord) and min(floor+1, n_in-1) with
    weight coord - floor.  Returned as a dense [n_out, n_in] matrix, one row per output pixel."""
    R = np.zeros((n_out, n_in), np.float64)
    ratio = np.float32(n_in) / np.float32(n_out)
    for i in range(n_out):
        coord = np.float32(i) * ratio
        base = int(math.floor(float(coord)))
        nxt = base + 1 if base + 1 <= n_in - 1 else n_in - 1
        frac = float(np.float32(coord - np.float32(base)))
        R[i, base] += 1.0 - frac
        R[i, nxt] += frac
    return torch.tensor(R, dtype=dtype)


# the `if OS == 8 ... else ...` of deeplabv3p.py:273-282: (entry_block3_stride, middle_block_rate, exit_block_rates,
# atrous_rates)
_XCEPTION_OS = {8: (1, 2, (2, 4), (12, 24, 36)), 16: (2, 1, (1, 2), (6, 12, 18))}

# MobileNetV2 body as the reference calls _inverted_res_block (deeplabv3p.py:327-367): one row per call,
# (filters, stride, expansion, block_id, skip_connection, rate); alpha = 1
_MNV2_CALLS = (
    (16, 1, 1, 0, False, 1),
    (24, 2, 6, 1, False, 1), (24, 1, 6, 2, True, 1),
    (32, 2, 6, 3, False, 1), (32, 1, 6, 4, True, 1), (32, 1, 6, 5, True, 1),
    (64, 1, 6, 6, False, 1), (64, 1, 6, 7, True, 2), (64, 1, 6, 8, True, 2), (64, 1, 6, 9, True, 2),
    (96, 1, 6, 10, False, 2), (96, 1, 6, 11, True, 2), (96, 1, 6, 12, True, 2),
    (160, 1, 6, 13, False, 2), (160, 1, 6, 14, True, 4), (160, 1, 6, 15, True, 4),
    (320, 1, 6, 16, False, 4),
)


class Ref:
    def __init__(self, params, training, dropout_mask=None, bn_frozen=False, dtype=torch.float32):
        self.np_params = params
        self.t = {}
        for k, v in params.items():
            t = torch.tensor(np.asarray(v), dtype=dtype)
            if not k.split("/")[-1].startswith("moving"):
                t.requires_grad_(True)
            self.t[k] = t
        self.training = training
        self.dropout_mask = dropout_mask
        self.bn_frozen = bn_frozen
        self.dtype = dtype
        self.batch_stats = {}  # BN layer name -> (batch mean, biased batch variance) of the last training forward
        self.bn_meta = {}      # BN layer name -> (epsilon, values per channel) of that forward
        self.record = None     # diagnostics (tools/r5/xception_layer_distance.py): BN layer name -> its output, NHWC numpy

    # x is NCHW throughout
    def conv(self, x, name, k=1, stride=1, same=True, bias=False):
        w = self.t[name + "/kernel:0"].permute(3, 2, 0, 1)  # HWIO -> OIHW
        H, W = x.shape[2], x.shape[3]
        if k == 1 and stride > 1:
            pads = (0, 0, 0, 0)
        else:
            fn = _same if same else _explicit
            pt, pb = fn(H, k, stride, 1)
            pl, pr = fn(W, k, stride, 1)
            pads = (pl, pr, pt, pb)
        x = F.pad(x, pads)
        b = self.t[name + "/bias:0"] if bias else None
        return F.conv2d(x, w, b, stride=stride)

    def dw(self, x, name, stride=1, rate=1, same=True):
        w = self.t[name + "/depthwise_kernel:0"].permute(2, 3, 0, 1)  # (3,3,C,1) -> (C,1,3,3)
        H, W = x.shape[2], x.shape[3]
        fn = _same if same else _explicit
        pt, pb = fn(H, 3, stride, rate)
        pl, pr = fn(W, 3, stride, rate)
        x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, w, None, stride=stride, dilation=rate, groups=x.shape[1])

    def bn(self, x, name, eps=1e-3):
        g, b = self.t[name + "/gamma:0"], self.t[name + "/beta:0"]
        mm, mv = self.t[name + "/moving_mean:0"], self.t[name + "/moving_variance:0"]
        if self.training and not self.bn_frozen:
            with torch.no_grad():
                self.batch_stats[name] = (x.mean(dim=(0, 2, 3)).numpy(), x.var(dim=(0, 2, 3), unbiased=False).numpy())
                self.bn_meta[name] = (eps, x.shape[0] * x.shape[2] * x.shape[3])
            return F.batch_norm(x, None, None, g, b, True, 0.0, _fused_eps(eps))
        y = F.batch_norm(x, mm, mv, g, b, False, 0.0, _fused_eps(eps))
        if self.record is not None:
            self.record[name] = y.detach().permute(0, 2, 3, 1).numpy().copy()
        return y

    def resize(self, x, Ho, Wo):
        """separable form out = Ry . x . Rx^T of the legacy bilinear resize (columns first, like TF's top/bottom lerp)"""
        Hi, Wi = x.shape[2], x.shape[3]
        Ry, Rx = _resize_matrix(Ho, Hi, x.dtype), _resize_matrix(Wo, Wi, x.dtype)
        cols = torch.einsum("nchw,xw->nchx", x, Rx)
        return torch.einsum("yh,nchx->ncyx", Ry, cols)

    def sepconv(self, x, prefix, stride=1, rate=1, depth_activation=False, eps=1e-3):
        if not depth_activation:
            x = F.relu(x)
        x = self.bn(self.dw(x, prefix + "_depthwise", stride, rate, same=(stride == 1)), prefix + "_depthwise_BN", eps)
        if depth_activation:
            x = F.relu(x)
        x = self.bn(self.conv(x, prefix + "_pointwise"), prefix + "_pointwise_BN", eps)
        if depth_activation:
            x = F.relu(x)
        return x

    def xblock(self, x, prefix, skip_type, stride, rate=1, depth_activation=False, return_skip=False):
        res, skip = x, None
        for i in range(3):
            res = self.sepconv(res, prefix + "_separable_conv%d" % (i + 1), stride if i == 2 else 1, rate,
                               depth_activation)
            if i == 1:
                skip = res
        if skip_type == "conv":
            sc = self.bn(self.conv(x, prefix + "_shortcut", 1, stride), prefix + "_shortcut_BN")
            out = res + sc
        elif skip_type == "sum":
            out = res + x
        else:
            out = res
        return (out, skip) if return_skip else out

    def features(self, x, backbone, input_shape, OS):
        H, W = input_shape[0], input_shape[1]
        x = x / 127.5 - 1.0
        skip1 = None
        if backbone == "xception":
            e3, mr, er, atrous = _XCEPTION_OS[8 if OS == 8 else 16]
            x = F.relu(self.bn(self.conv(x, "entry_flow_conv1_1", 3, 2), "entry_flow_conv1_1_BN"))
            x = F.relu(self.bn(self.conv(x, "entry_flow_conv1_2", 3, 1), "entry_flow_conv1_2_BN"))
            x = self.xblock(x, "entry_flow_block1", "conv", 2)
            x, skip1 = self.xblock(x, "entry_flow_block2", "conv", 2, return_skip=True)
            x = self.xblock(x, "entry_flow_block3", "conv", e3)
            for i in range(16):
                x = self.xblock(x, "middle_flow_unit_%d" % (i + 1), "sum", 1, mr)
            x = self.xblock(x, "exit_flow_block1", "conv", 1, er[0])
            x = self.xblock(x, "exit_flow_block2", "none", 1, er[1], depth_activation=True)
        else:
            OS = 8
            x = F.relu6(self.bn(self.conv(x, "Conv", 3, 2), "Conv_BN"))
            for filters, stride, expansion, bid, skip, rate in _MNV2_CALLS:
                inp = x
                prefix = "expanded_conv_%d_" % bid if bid else "expanded_conv_"
                if bid:
                    x = F.relu6(self.bn(self.conv(x, prefix + "expand"), prefix + "expand_BN"))
                x = F.relu6(self.bn(self.dw(x, prefix + "depthwise", stride, rate), prefix + "depthwise_BN"))
                x = self.bn(self.conv(x, prefix + "project"), prefix + "project_BN")
                if skip:
                    x = inp + x
        fh, fw = math.ceil(H / OS), math.ceil(W / OS)
        b4 = x.mean(dim=(2, 3), keepdim=True)
        b4 = F.relu(self.bn(self.conv(b4, "image_pooling"), "image_pooling_BN", 1e-5))
        b4 = self.resize(b4, fh, fw)
        b0 = F.relu(self.bn(self.conv(x, "aspp0"), "aspp0_BN", 1e-5))
        if backbone == "xception":
            bs = [self.sepconv(x, "aspp%d" % (i + 1), 1, atrous[i], True, 1e-5) for i in range(3)]
            x = torch.cat([b4, b0] + bs, dim=1)
        else:
            x = torch.cat([b4, b0], dim=1)
        x = F.relu(self.bn(self.conv(x, "concat_projection"), "concat_projection_BN", 1e-5))
        if self.training and self.dropout_mask is not None:
            m = torch.tensor(self.dropout_mask, dtype=x.dtype).permute(0, 3, 1, 2)
            x = x * m * (1.0 / 0.9)
        if backbone == "xception":
            x = self.resize(x, math.ceil(H / 4), math.ceil(W / 4))
            d = F.relu(self.bn(self.conv(skip1, "feature_projection0"), "feature_projection0_BN", 1e-5))
            x = torch.cat([x, d], dim=1)
            x = self.sepconv(x, "decoder_conv0", 1, 1, True, 1e-5)
            x = self.sepconv(x, "decoder_conv1", 1, 1, True, 1e-5)
        return x

    def logits(self, x_nhwc, backbone="mobilenetv2", input_shape=(512, 512, 3), classes=21, OS=16, head="deeplab",
               subpixel_name="subpixel_1"):
        x = torch.tensor(np.asarray(x_nhwc), dtype=self.dtype).permute(0, 3, 1, 2)
        f = self.features(x, backbone, input_shape, OS)
        H, W = input_shape[0], input_shape[1]
        if head == "subpixel":
            r = 4 if backbone == "xception" else 8
            y = self.conv(f, subpixel_name, bias=True)  # [N, C*r*r, a, b]
            N, c, a, b = y.shape
            co = c // (r * r)
            # out[n, ia*r+q, ib*r+p, ch] = I[n, ia, ib, ch*r*r + p*r + q]  (subpixel.py:77-88)
            y = y.reshape(N, co, r, r, a, b)           # [n, ch, p, q, ia, ib]
            y = y.permute(0, 1, 4, 3, 5, 2)            # [n, ch, ia, q, ib, p]
            y = y.reshape(N, co, a * r, b * r)
        else:
            name = "conv_upsample" if head == "original" else (
                "logits_semantic" if classes == 21 else "custom_logits_semantic")
            y = self.resize(self.conv(f, name, bias=True), H, W)
        return y.permute(0, 2, 3, 1)  # NHWC

    def loss(self, logits, labels, weights):
        B, H, W, C = logits.shape
        lg = logits.reshape(B, H * W, C)
        t = torch.tensor(np.asarray(labels).reshape(B, H * W)).long()
        w = torch.tensor(np.asarray(weights).reshape(B, H * W), dtype=self.dtype)
        onehot = F.one_hot(t, C + 1)[..., :C].to(self.dtype)  # utils.py:129
        p = torch.softmax(lg, dim=-1)
        q = p / p.sum(dim=-1, keepdim=True)
        # tf.clip_by_value: identity gradient inside [1e-7, 1-1e-7] (bounds included), zero outside — torch.clamp's own
        q = q.clamp(1e-7, 1 - 1e-7)
        l = -(onehot * torch.log(q)).sum(dim=-1)
        nnz = max(float((w != 0).sum()), 1.0)
        return (l * w).sum() / nnz


def train_grads(params, x, labels, weights, dropout_mask=None, bn_frozen=False, dtype=torch.float32, **kw):
    ref = Ref(params, True, dropout_mask, bn_frozen, dtype)
    logits = ref.logits(x, **kw)
    loss = ref.loss(logits, labels, weights)
    loss.backward()
    grads = {k: (None if t.grad is None else t.grad.numpy()) for k, t in ref.t.items() if t.requires_grad}
    return float(loss.detach()), grads, logits.detach().numpy()


def infer_logits(params, x, dtype=torch.float32, **kw):
    with torch.no_grad():
        return Ref(params, False, dtype=dtype).logits(x, **kw).numpy()


def calibrate_bn(params, x, dtype=torch.float64, **kw):
    """moving statistics := batch statistics of one training-mode forward on x (keeps the activations of a randomly
    initialised net O(1) through 50-140 layers); the full-size counterpart of dl3_oracle.calibrate_bn."""
    with torch.no_grad():
        ref = Ref(params, True, dtype=dtype)
        ref.logits(x, **kw)
    out = dict(params)
    for name, (m, v) in ref.batch_stats.items():
        out[name + "/moving_mean:0"] = m.astype(np.float32)
        out[name + "/moving_variance:0"] = v.astype(np.float32)
    return out


def _bn_momentum(name):
    """BatchNormalization momentum by layer: 0.999 where the reference passes it (the MobileNetV2 backbone,
    deeplabv3p.py:178,189,197,322-323), the Keras default 0.99 everywhere else"""
    return 0.999 if (name == "Conv_BN" or name.startswith("expanded_conv")) else 0.99


def train_steps(params, batches, opt=None, bn_frozen=False, dtype=torch.float64, **kw):
    """Independent restatement of k x train_on_batch with the notebook's optimizer (segmentation.ipynb json 107:
    Adam(lr=7e-4, epsilon=1e-8, decay=1e-6); Keras 2.2.4 Adam.get_updates) for tests/: torch autograd gradients, the
    Adam arithmetic on torch tensors, the Keras 2.2.4 / TF 1.13 moving-statistics update (Bessel x n/(n-(1+eps))).
    Shares no code with oracle/dl3_oracle.py.  Returns (losses, params after the last step)."""
    o = dict(lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6)
    o.update(opt or {})
    p = {k: np.asarray(v) for k, v in params.items()}
    mom1, mom2, losses = {}, {}, []
    for it, (x, y, w) in enumerate(batches):
        ref = Ref(p, True, None, bn_frozen, dtype)
        loss = ref.loss(ref.logits(x, **kw), y, w)
        loss.backward()
        losses.append(float(loss.detach()))
        lr = o["lr"] / (1.0 + o["decay"] * it) if o["decay"] > 0 else o["lr"]
        t = it + 1
        lr_t = lr * math.sqrt(1.0 - o["beta_2"] ** t) / (1.0 - o["beta_1"] ** t)
        new = dict(p)
        with torch.no_grad():
            for name, wt in ref.t.items():
                if wt.grad is None:
                    continue
                m = mom1.get(name, torch.zeros_like(wt))
                v = mom2.get(name, torch.zeros_like(wt))
                m = o["beta_1"] * m + (1.0 - o["beta_1"]) * wt.grad
                v = o["beta_2"] * v + (1.0 - o["beta_2"]) * wt.grad * wt.grad
                mom1[name], mom2[name] = m, v
                new[name] = (wt - lr_t * m / (torch.sqrt(v) + o["epsilon"])).numpy()
        for name, (bm, bv) in ref.batch_stats.items():
            eps, n = ref.bn_meta[name]
            unb = bv * (n / (n - 1.0)) if n > 1 else bv
            if n > 1.0 + eps:
                unb = unb * (n / (n - (1.0 + eps)))
            mo = _bn_momentum(name)
            new[name + "/moving_mean:0"] = mo * p[name + "/moving_mean:0"] + (1.0 - mo) * bm
            new[name + "/moving_variance:0"] = mo * p[name + "/moving_variance:0"] + (1.0 - mo) * unb
        p = new
    return losses, p
