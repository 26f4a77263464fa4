"""Host-side mirror of the small slice of the Keras functional API that the reference's hot path
touches (deeplabv3p.py:29-40 imports; used at :47-206, :260-452; utils.py:181-198).

Pure Python + numpy: layers only DECLARE the graph (names, hyper-parameters, weight shapes and
host master copies of the weights).  All arithmetic happens in libdl3.so through engine.py.
What callers of the reference touch is reproduced: `Model.input/.output/.layers/.name`,
`layer.name/.output/.trainable/.get_weights()/.set_weights()`, `Model(inputs, outputs)` re-wiring
(utils.py:181,:193,:198), Keras 2.2.4 layer ordering (`layers[-5]`) and auto-naming.
"""
import collections
import math
import os

import numpy as np

_uids = collections.defaultdict(int)
_rng = np.random.default_rng(0)


def clear_session(seed=0):
    """keras.backend.clear_session(): reset auto-naming counters (and the weight-init RNG)."""
    global _rng
    _uids.clear()
    _rng = np.random.default_rng(seed)


def set_seed(seed):
    global _rng
    _rng = np.random.default_rng(seed)


def _auto_name(prefix):
    _uids[prefix] += 1
    return "%s_%d" % (prefix, _uids[prefix])


def glorot_uniform(shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return _rng.uniform(-lim, lim, shape).astype(np.float32)


def glorot_normal(shape, fan_in, fan_out):
    """tf.glorot_normal_initializer (default of icnr_weights, subpixel.py:9): truncated normal,
    stddev = sqrt(2/(fan_in+fan_out))/.87962566 [TF-semantics]; the RNG stream itself is not
    reproducible against TF, only the distribution."""
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    v = _rng.normal(0.0, std, shape)
    bad = np.abs(v) > 2 * std
    while bad.any():
        v[bad] = _rng.normal(0.0, std, int(bad.sum()))
        bad = np.abs(v) > 2 * std
    return v.astype(np.float32)


class KTensor:
    """Symbolic tensor: static shape without the batch axis, and the layer that produced it."""

    def __init__(self, shape, layer):
        self.shape = tuple(shape)
        self.layer = layer
        self._keras_shape = (None,) + self.shape  # read by the reference at deeplabv3p.py:168

    def __repr__(self):
        return "<KTensor %s from %s>" % (self.shape, self.layer.name)


class Layer:
    kind = "Layer"
    prefix = "layer"

    def __init__(self, name=None, **cfg):
        self.name = name or _auto_name(self.prefix)
        self.cfg = cfg
        self.trainable = True
        self.inbound = []
        self.output = None
        self.weights = collections.OrderedDict()  # "<layer>/<var>:0" -> np.ndarray (host master copy)
        self._engine = None

    # Keras: layer(x) / layer([a, b])
    def __call__(self, x):
        ins = list(x) if isinstance(x, (list, tuple)) else [x]
        assert self.output is None, "layer %s called twice (shared layers are not on the path)" % self.name
        self.inbound = ins
        self.output = KTensor(self.compute_output_shape([t.shape for t in ins]), self)
        self.build([t.shape for t in ins])
        return self.output

    @property
    def input(self):
        return self.inbound[0] if len(self.inbound) == 1 else self.inbound

    def build(self, in_shapes):
        pass

    def compute_output_shape(self, in_shapes):
        return in_shapes[0]

    def weight_names(self):
        return list(self.weights.keys())

    def get_weights(self):
        if self._engine is not None:
            self._engine.sync_layer_to_host(self)
        return [w.copy() for w in self.weights.values()]

    def set_weights(self, ws):
        names = list(self.weights.keys())
        if len(ws) != len(names):
            raise ValueError("layer %s expects %d weight arrays, got %d" % (self.name, len(names), len(ws)))
        for n, w in zip(names, ws):
            w = np.asarray(w, np.float32)
            if w.shape != self.weights[n].shape:
                raise ValueError("layer %s weight %s: shape %s != %s" % (self.name, n, w.shape, self.weights[n].shape))
            self.weights[n] = w.copy()
        if self._engine is not None:
            self._engine.sync_layer_to_device(self)

    def count_params(self):
        return int(sum(w.size for w in self.weights.values()))


class InputLayer(Layer):
    kind = "InputLayer"
    prefix = "input"


def Input(shape=None, tensor=None, name=None):
    """keras.layers.Input (deeplabv3p.py:261,:264).  `tensor` may be a numpy/torch array whose
    trailing dims give the shape; it is remembered as the model's default feed."""
    lyr = InputLayer(name=name)
    if shape is None and tensor is not None:
        shape = tuple(tensor.shape[1:])
    lyr.output = KTensor(shape, lyr)
    lyr.cfg["tensor"] = tensor
    return lyr.output


def _same_out(size, stride):
    return -(-size // stride)


class Conv2D(Layer):
    """keras.layers.Conv2D restricted to what the path uses: k in {1,3}, dilation 1, groups 1."""
    kind = "Conv2D"
    prefix = "conv2d"

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, dilation_rate=(1, 1),
                 activation=None, name=None, kernel_initializer="glorot_uniform", **kw):
        super().__init__(name=name)
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        s = strides[0] if isinstance(strides, (tuple, list)) else strides
        r = dilation_rate[0] if isinstance(dilation_rate, (tuple, list)) else dilation_rate
        assert activation is None and r == 1
        self.cfg.update(filters=int(filters), k=int(k), stride=int(s), padding=padding, use_bias=bool(use_bias), rate=1)

    def compute_output_shape(self, in_shapes):
        H, W, _ = in_shapes[0]
        c = self.cfg
        if c["padding"] == "same":
            return (_same_out(H, c["stride"]), _same_out(W, c["stride"]), c["filters"])
        return ((H - c["k"]) // c["stride"] + 1, (W - c["k"]) // c["stride"] + 1, c["filters"])

    def build(self, in_shapes):
        cin, c = in_shapes[0][2], self.cfg
        k, f = c["k"], c["filters"]
        self.weights[self.name + "/kernel:0"] = glorot_uniform((k, k, cin, f), k * k * cin, k * k * f)
        if c["use_bias"]:
            self.weights[self.name + "/bias:0"] = np.zeros((f,), np.float32)


class DepthwiseConv2D(Layer):
    kind = "DepthwiseConv2D"
    prefix = "depthwise_conv2d"

    def __init__(self, kernel_size, strides=(1, 1), padding="valid", use_bias=True, dilation_rate=(1, 1),
                 activation=None, name=None, **kw):
        super().__init__(name=name)
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        s = strides[0] if isinstance(strides, (tuple, list)) else strides
        r = dilation_rate[0] if isinstance(dilation_rate, (tuple, list)) else dilation_rate
        assert k == 3 and not use_bias and activation is None
        self.cfg.update(k=3, stride=int(s), padding=padding, rate=int(r))

    def compute_output_shape(self, in_shapes):
        H, W, C = in_shapes[0]
        c = self.cfg
        if c["padding"] == "same":
            return (_same_out(H, c["stride"]), _same_out(W, c["stride"]), C)
        keff = (c["k"] - 1) * c["rate"] + 1
        return ((H - keff) // c["stride"] + 1, (W - keff) // c["stride"] + 1, C)

    def build(self, in_shapes):
        C = in_shapes[0][2]
        # Keras DepthwiseConv2D glorot fans: fan_in = kh*kw*C, fan_out = kh*kw*depth_multiplier
        self.weights[self.name + "/depthwise_kernel:0"] = glorot_uniform((3, 3, C, 1), 9 * C, 9)


class BatchNormalization(Layer):
    kind = "BatchNormalization"
    prefix = "batch_normalization"

    def __init__(self, epsilon=1e-3, momentum=0.99, name=None, **kw):
        super().__init__(name=name)
        self.cfg.update(eps=float(epsilon), momentum=float(momentum))

    def build(self, in_shapes):
        C = in_shapes[0][-1]
        n = self.name
        self.weights[n + "/gamma:0"] = np.ones((C,), np.float32)
        self.weights[n + "/beta:0"] = np.zeros((C,), np.float32)
        self.weights[n + "/moving_mean:0"] = np.zeros((C,), np.float32)
        self.weights[n + "/moving_variance:0"] = np.ones((C,), np.float32)


class Activation(Layer):
    """'relu', 'relu6' (the reference's Lambda(relu(x, max_value=6.)), deeplabv3p.py:181) or 'softmax'."""
    kind = "Activation"
    prefix = "activation"

    def __init__(self, fn, name=None):
        super().__init__(name=name)
        assert fn in ("relu", "relu6", "softmax")
        self.cfg["fn"] = fn


class ReLU6(Activation):
    """Lambda(lambda x: relu(x, max_value=6.)) — a Lambda layer in the reference, hence lambda_N auto-names."""
    prefix = "lambda"

    def __init__(self, name=None):
        super().__init__("relu6", name=name)


class Prescale(Layer):
    """Lambda(lambda x: x/127.5 - 1) (deeplabv3p.py:270)"""
    kind = "Prescale"
    prefix = "lambda"


class ResizeBilinear(Layer):
    """Lambda(K.tf.image.resize_bilinear(x, size)) (deeplabv3p.py:382,:418,:439; utils.py:190)"""
    kind = "ResizeBilinear"
    prefix = "lambda"

    def __init__(self, size, name=None):
        super().__init__(name=name)
        self.cfg["size"] = (int(size[0]), int(size[1]))

    def compute_output_shape(self, in_shapes):
        return self.cfg["size"] + (in_shapes[0][2],)


class Add(Layer):
    kind = "Add"
    prefix = "add"

    def compute_output_shape(self, in_shapes):
        assert in_shapes[0] == in_shapes[1], in_shapes
        return in_shapes[0]


class Concatenate(Layer):
    kind = "Concatenate"
    prefix = "concatenate"

    def compute_output_shape(self, in_shapes):
        assert all(s[:2] == in_shapes[0][:2] for s in in_shapes), in_shapes
        return in_shapes[0][:2] + (sum(s[2] for s in in_shapes),)


class AveragePooling2D(Layer):
    kind = "AveragePooling2D"
    prefix = "average_pooling2d"

    def __init__(self, pool_size, name=None):
        super().__init__(name=name)
        self.cfg["pool"] = (int(pool_size[0]), int(pool_size[1]))

    def compute_output_shape(self, in_shapes):
        H, W, C = in_shapes[0]
        ph, pw = self.cfg["pool"]
        if (H // ph, W // pw) != (1, 1):
            raise NotImplementedError("only the global pooling of deeplabv3p.py:375 is on the path")
        return (1, 1, C)


class ZeroPadding2D(Layer):
    kind = "ZeroPadding2D"
    prefix = "zero_padding2d"

    def __init__(self, padding, name=None):
        super().__init__(name=name)
        # Keras reads a 2-tuple of ints as symmetric (height, width) padding (deeplabv3p.py:68,:110)
        ph, pw = padding
        self.cfg["pad"] = ((int(ph), int(ph)), (int(pw), int(pw)))

    def compute_output_shape(self, in_shapes):
        H, W, C = in_shapes[0]
        (pt, pb), (pl, pr) = self.cfg["pad"]
        return (H + pt + pb, W + pl + pr, C)


class Dropout(Layer):
    kind = "Dropout"
    prefix = "dropout"

    def __init__(self, rate, name=None):
        super().__init__(name=name)
        self.cfg["rate"] = float(rate)


class Reshape(Layer):
    kind = "Reshape"
    prefix = "reshape"

    def __init__(self, target_shape, name=None):
        super().__init__(name=name)
        self.cfg["target"] = tuple(target_shape)

    def compute_output_shape(self, in_shapes):
        n = int(np.prod(in_shapes[0]))
        tgt = list(self.cfg["target"])
        if -1 in tgt:
            known = int(np.prod([t for t in tgt if t != -1]))
            tgt[tgt.index(-1)] = n // known
        assert int(np.prod(tgt)) == n, (tgt, in_shapes)
        return tuple(tgt)


# --------------------------------------------------------------------------------------


class Model:
    """keras.models.Model(inputs, outputs, name): the sub-graph between `inputs` and `outputs`."""

    def __init__(self, inputs, outputs, name=None):
        self.name = name or _auto_name("model")
        self.input = inputs
        self.output = outputs
        self._collect()
        self._engines = {}
        self._compiled = None
        self._dp = None          # parallel.DataParallel once distribute() was called
        self._dp_synced = False
        self._train_eng = None   # the engine that holds the live Adam moments / iteration / dropout step

    # -- graph -----------------------------------------------------------------------
    def _collect(self):
        """Keras 2.2.4 Network._init_graph_network ordering: DFS from the output
        assigns visit indices (pre-order, inputs in call order), depth = longest path to the
        output, layers sorted by decreasing depth then increasing visit index."""
        out_layer = self.output.layer
        in_layer = self.input.layer
        index, depth = {}, {}
        finished = set()

        def visit(start):
            # iterative form of Keras' recursive build_map: a layer gets its index when it is first
            # ENTERED (pre-order); inbound layers are entered in call order; finished layers are skipped
            stack = [(start, 0)]
            while stack:
                lyr, i = stack.pop()
                if i == 0:
                    if lyr in finished:
                        continue
                    if lyr not in index:
                        index[lyr] = len(index)
                ins = [] if lyr is in_layer else lyr.inbound
                if i < len(ins):
                    stack.append((lyr, i + 1))
                    stack.append((ins[i].layer, 0))
                else:
                    finished.add(lyr)

        visit(out_layer)
        if in_layer not in index:
            raise ValueError("graph disconnected: the input is not an ancestor of the output")
        # topological (creation) order: every layer's inputs precede it
        topo = []
        seen = set()

        def topo_visit(layer):
            stack = [(layer, 0)]
            while stack:
                lyr, i = stack.pop()
                if lyr in seen:
                    continue
                ins = lyr.inbound if lyr is not in_layer else []
                if i < len(ins):
                    stack.append((lyr, i + 1))
                    if ins[i].layer not in seen:
                        stack.append((ins[i].layer, 0))
                else:
                    seen.add(lyr)
                    topo.append(lyr)

        topo_visit(out_layer)
        self._topo = topo
        # depth = longest path to the output, relaxed in reverse topological order
        for lyr in topo:
            depth[lyr] = 0
        for lyr in reversed(topo):
            ins = lyr.inbound if lyr is not in_layer else []
            for t in ins:
                depth[t.layer] = max(depth[t.layer], depth[lyr] + 1)
        self.layers = sorted(topo, key=lambda l: (-depth[l], index[l]))
        self._by_name = {l.name: l for l in self.layers}

    def get_layer(self, name=None, index=None):
        if index is not None:
            return self.layers[index]
        return self._by_name[name]

    @property
    def weights(self):
        return [n for l in self.layers for n in l.weights]

    def count_params(self):
        return sum(l.count_params() for l in self.layers)

    def trainable_count(self):
        n = 0
        for l in self.layers:
            if not l.trainable:
                continue
            for k, w in l.weights.items():
                if "/moving_" not in k:
                    n += w.size
        return int(n)

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        for l in self.layers:
            print_fn("%-48s %-22s %-18s %9d" % (l.name, l.kind, l.output.shape, l.count_params()))
        print_fn("Total params: %d   Trainable params: %d" % (self.count_params(), self.trainable_count()))

    # -- weights -----------------------------------------------------------------------
    def get_weights(self):
        return [w for l in self.layers for w in l.get_weights()]

    def set_weights(self, ws):
        i = 0
        for l in self.layers:
            n = len(l.weights)
            if n:
                l.set_weights(ws[i:i + n])
                i += n

    def save_weights(self, path):
        from . import h5io
        h5io.save_weights(self, path)

    def load_weights(self, path, by_name=False):
        from . import h5io
        h5io.load_weights(self, path, by_name=by_name)

    # -- execution (delegated to the HIP engine) ----------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, sample_weight_mode=None, **kw):
        """keras Model.compile as used by the notebook (cell 2): remembers the optimizer
        hyper-parameters; the loss on the path is always sparse_crossentropy_ignoring_last_label
        with temporal sample weights (utils.py:127-130)."""
        from .optimizers import as_adam_dict
        # optimizer: None (the notebook's Adam(lr=7e-4, epsilon=1e-8, decay=1e-6)), a dict of overrides, 'adam', or an
        # optimizers.Adam / any Keras-style Adam exposing get_config(); anything else raises here, not at the first step
        self._compiled = dict(optimizer=as_adam_dict(optimizer), optimizer_object=optimizer, loss=loss, metrics=metrics,
                              sample_weight_mode=sample_weight_mode)
        # Keras 2.2.4 collects the weights the optimizer updates HERE (`_collected_trainable_weights`), and the layers'
        # update ops (BatchNormalization moving statistics) when the train function is first built — the first
        # train_on_batch / fit after compile().  `layer.trainable` flipped in between (segmentation.ipynb: compile in
        # cell 2, the fine-tuning loop in cell 5, no recompile) therefore changes which moving statistics move but NOT
        # which weights train: Keras warns about the discrepancy and trains them all.  Same here.
        self._compile_gen = getattr(self, "_compile_gen", 0) + 1
        self._collected_trainable = {l.name: bool(l.trainable) for l in self.layers}
        self._update_flags = None   # frozen at the first training step after this compile()
        # a new train function starts from fresh optimizer slots (Keras re-creates m / v in get_updates) — but
        # `optimizer.iterations` is a variable of the OPTIMIZER OBJECT: compiling again with the same Adam instance (the
        # notebook's fine-tuning recompile) keeps counting, so the lr decay and the bias-correction exponent continue
        # (ADVICE r5).  A new object, a dict or a string starts at 0.
        prev, prev_obj = self._train_eng, getattr(self, "_opt_object", None)
        keep = prev is not None and optimizer is not None and optimizer is prev_obj and not isinstance(optimizer, (dict, str))
        self._carry_iterations = (int(prev.iteration), prev.drop_step.clone()) if keep else None
        self._opt_object = optimizer
        self._train_eng = None

    def distribute(self, dp=None, strict=True):
        """Per-image data parallelism for train_on_batch / fit: this process is one of WORLD_SIZE (one per GPU, launched
        by torch.distributed.run); every call hands over the GLOBAL batch (or, with global_batch=False, this rank's own
        shard), each rank trains on its contiguous shard and the gradients are summed with one RCCL all-reduce
        (parallel.DataParallel).  Stands in for keras.utils.multi_gpu_model (utils.py:209-211), whose towers likewise keep
        per-replica BatchNorm statistics.
        strict (default): an RCCL communicator that cannot be brought up RAISES — a silent fall-back to gloo stages every
        gradient exchange through host memory (10-100x slower); strict=False, or DL3_DIST_BACKEND=gloo asked for explicitly,
        keeps the warning-and-gloo behaviour the functional tests use."""
        from .parallel import DataParallel
        self._dp = dp if dp is not None else DataParallel(strict=strict)
        self._dp_synced = False
        return self

    MAX_ENGINES = int(os.environ.get("DL3_MAX_ENGINES", "4"))

    def _training_flags(self):
        """({layer: does the optimizer update its weights}, {layer: do its update ops run}, cache key) for a training engine.
        Compiled model: the Keras 2.2.4 rule (see compile()).  Never compiled: the live flags, and the key carries them, so a
        plan lowered for one set of flags is never replayed for another."""
        live = {l.name: bool(l.trainable) for l in self.layers}
        if getattr(self, "_collected_trainable", None) is None:
            return live, live, ("live", tuple(sorted(n for n, t in live.items() if not t)))
        if self._update_flags is None:
            self._update_flags = live
            if live != self._collected_trainable:
                import warnings
                warnings.warn("dl3: discrepancy between trainable weights and collected trainable weights — "
                              "`layer.trainable` changed after compile(): as in Keras 2.2.4 the optimizer still updates the "
                              "weights collected at compile(); only the moving statistics of now-frozen BatchNormalization "
                              "layers stop moving.  Call compile() again after setting `trainable` to train the tail only.",
                              UserWarning, stacklevel=4)
        return self._collected_trainable, self._update_flags, ("compiled", self._compile_gen)

    def _engine(self, batch, training, **kw):
        """engine for (batch, mode); weights travel through the host master copies when the active engine changes, the
        optimizer state (Adam moments, iteration, dropout step) moves device-to-device between training engines, and
        the least recently used engines beyond MAX_ENGINES are dropped (each one owns a full activation arena)."""
        from .engine import Engine
        key = (int(batch), bool(training), tuple(sorted(kw.items())))
        if training:
            flags = self._training_flags()
            kw = dict(kw, opt_trainable=flags[0], bn_update=flags[1])
            key = key + (flags[2],)
        eng = self._engines.pop(key, None)
        active = getattr(self, "_active", None)
        if eng is not active and active is not None:
            active.sync_all_to_host()
        if eng is None:
            eng = Engine(self, batch=int(batch), training=training, **kw)
        elif eng is not active:
            eng.sync_all_to_device()
        self._engines[key] = eng  # most recently used last
        if training:
            prev = self._train_eng
            if prev is not None and prev is not eng:
                eng.adopt_optimizer_state(prev)  # e.g. the last, smaller batch of an epoch keeps the same Adam
            elif prev is None and getattr(self, "_carry_iterations", None) is not None:
                # first train function after a compile() that re-used the optimizer object: fresh moments, same clock
                it, drop = self._carry_iterations
                eng.iteration = it
                eng.adam_m.zero_()
                eng.adam_v.zero_()
                eng.drop_step.copy_(drop)
                self._carry_iterations = None
            self._train_eng = eng
        while len(self._engines) > self.MAX_ENGINES:
            k0 = next(iter(self._engines))
            old = self._engines[k0]
            if old is eng or old is self._train_eng:
                self._engines[k0] = self._engines.pop(k0)  # keep; move to the young end
                if all(e is eng or e is self._train_eng for e in self._engines.values()):
                    break
                continue
            del self._engines[k0]
        self._active = eng
        eng.activate()
        return eng

    def predict(self, x, batch_size=32, verbose=0):
        if not hasattr(x, "data_ptr") and not (isinstance(x, np.ndarray) and x.dtype == np.uint8):
            x = np.asarray(x, np.float32)
        n = x.shape[0]
        bs = min(int(batch_size), n)
        outs = []
        for i in range(0, n, bs):
            xb = x[i:i + bs]
            eng = self._engine(xb.shape[0], False)
            outs.append(eng.predict(xb))
        return np.concatenate(outs, axis=0)

    def predict_mask(self, x, batch_size=32):
        """np.argmax(model.predict(x), -1) (notebook cell 9) without shipping the probabilities to the host: the argmax
        runs on the device (dl3_argmax) and only the int32 masks [B,H,W] cross PCIe — 1 MB instead of 22 MB per
        512x512x21 image.  Not part of the reference's Model API."""
        if not hasattr(x, "data_ptr") and not (isinstance(x, np.ndarray) and x.dtype == np.uint8):
            x = np.asarray(x, np.float32)
        n = x.shape[0]
        bs = min(int(batch_size), n)
        outs = []
        for i in range(0, n, bs):
            xb = x[i:i + bs]
            eng = self._engine(xb.shape[0], False)
            eng.set_input(xb)
            eng.forward()
            outs.append(eng.argmax())
        return np.concatenate(outs, axis=0)

    def evaluate(self, x, y, batch_size=32, sample_weight=None, verbose=0):
        """keras Model.evaluate for the notebook's metrics (cell 2: metrics=[Jaccard, sparse_accuracy_ignoring_last_label]):
        returns [loss, Jaccard, accuracy].  The argmax mask and the per-image/per-class pixel counts are produced on
        the device (dl3_argmax, dl3_seg_counts); the metric ratios (utils.py:132-157) and the loss (utils.py:127-130,
        Keras weighted mean) are evaluated on the host from those counts / the probabilities."""
        from . import utils as U
        if not hasattr(x, "data_ptr") and not (isinstance(x, np.ndarray) and x.dtype == np.uint8):
            x = np.asarray(x, np.float32)
        y = np.asarray(y)
        n = x.shape[0]
        bs = min(int(batch_size), n)
        counts, num, den = [], 0.0, 0.0
        for i in range(0, n, bs):
            xb, yb = x[i:i + bs], y[i:i + bs]
            eng = self._engine(xb.shape[0], False)
            probs = eng.predict(xb)
            counts.append(eng.seg_counts(yb))
            C = probs.shape[-1]
            probs = probs.reshape(xb.shape[0], -1, C)
            yb = yb.reshape(xb.shape[0], -1, 1)
            # no sample weights: Keras takes the plain mean over all B*HW pixels (void rows contribute 0 through the
            # one-hot, utils.py:129); with weights: sum(l*w) / count(w != 0)
            w = np.ones(yb.shape[:2], np.float64) if sample_weight is None else \
                np.asarray(sample_weight[i:i + bs], np.float64).reshape(xb.shape[0], -1)
            ell = U.sparse_crossentropy_ignoring_last_label(yb, probs)
            num += float((ell * w).sum() / max((w != 0).mean(), 1e-30) / w.size) * xb.shape[0]
            den += xb.shape[0]
        counts = np.concatenate(counts, 0)
        return [num / den, U.Jaccard_from_counts(counts), U.accuracy_from_counts(counts)]

    def _dp_active(self):
        dp = self._dp
        return dp is not None and (dp.world > 1 or dp.comm is not None)

    def _dp_sync_weights(self, eng):
        """identical weights and moving statistics on every rank before the first data-parallel step"""
        if not self._dp_synced:
            self._dp.broadcast(eng.params)
            self._dp.broadcast(eng.state)
            eng.dirty = True
            self._dp_synced = True

    def train_on_batch(self, x, y, sample_weight=None, lazy_loss=False, global_batch=True, **engine_kw):
        """keras Model.train_on_batch; lazy_loss=True returns an engine.LazyLoss (float() reads it) instead of stalling
        the stream for one scalar every step — fit / fit_generator use it and read once per epoch.
        Under distribute(): global_batch=True (Keras' multi_gpu_model contract, utils.py:209-211) — every rank is handed
        the whole batch and keeps its contiguous shard; global_batch=False — (x, y, sample_weight) ARE this rank's shard
        (a generator sharded by rank: nothing redundant is built or copied on the host)."""
        dp = self._dp
        # the data-parallel step also runs for a communicator of ONE rank (dp.comm set): every launch of the N-rank step —
        # the hipGraph replay, the ONE all-reduce of the arena (gradients + the shard's count and loss sum) behind it, Adam
        # finishing the scale on the device — on one GPU
        multi = dp is not None and (dp.world > 1 or dp.comm is not None)
        if multi:
            n = x.shape[0]
            if global_batch and n < dp.world:
                raise ValueError("global batch of %d images cannot be split over %d ranks" % (n, dp.world))
            # a ragged batch (the last one of a Sequence) gives its remainder to the last rank, like the reference's
            # multi_gpu_model towers; the loss is normalised by the GLOBAL count(w != 0) (Engine.train_step)
            if global_batch:
                lo, hi = dp.shard(n)
                x, y = x[lo:hi], y[lo:hi]
                if sample_weight is not None:
                    sample_weight = sample_weight[lo:hi]
            engine_kw = dict(engine_kw, external_nnz=True)
        eng = self._engine(x.shape[0], True, **engine_kw)
        opt = (self._compiled or {}).get("optimizer") or {}
        if not multi:
            return eng.train_step(x, y, sample_weight, opt, lazy=lazy_loss)
        self._dp_sync_weights(eng)
        # sum_all(l*w) / count_all(w != 0), from the arena tail the one all-reduce summed: the same number on every rank
        return eng.train_step(x, y, sample_weight, opt, comm=dp, lazy=lazy_loss)

    _IGNORED_FIT_KW = ("workers", "use_multiprocessing", "max_queue_size", "shuffle", "initial_epoch")

    def _check_fit_kw(self, kw):
        import warnings
        for k, v in kw.items():
            if k in self._IGNORED_FIT_KW or v is None or (isinstance(v, (list, tuple)) and not v):
                continue
            warnings.warn("dl3: Model.fit/fit_generator does not implement %r (the training loop of the reference, "
                          "utils.py:216-254, is host-side control plane outside this package): it is IGNORED — drive "
                          "callbacks / validation from your own loop around train_on_batch / evaluate" % k,
                          RuntimeWarning, stacklevel=3)

    def fit(self, x, y, batch_size=16, epochs=1, sample_weight=None, verbose=0, **kw):
        """Minimal Model.fit (utils.py:244): plain epochs over (x, y) without shuffling or callbacks."""
        self._check_fit_kw(kw)
        hist = []
        n = x.shape[0]
        for _ in range(epochs):
            for i in range(0, n - batch_size + 1, batch_size):
                sw = None if sample_weight is None else sample_weight[i:i + batch_size]
                hist.append(self.train_on_batch(x[i:i + batch_size], y[i:i + batch_size], sw, lazy_loss=True))
        return [float(l) for l in hist]

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=0, device_feed=False, n_classes=None,
                      global_batch=True, **kw):
        """Minimal Model.fit_generator (utils.py:233): generator yields (X, Y, {'pred_mask': SW}) or (X, Y, SW).
        device_feed=True (not in Keras): the generator yields (uint8 images [B,H,W,3], raw label maps [B,H,W] uint8 / int32)
        — what cv2 decodes, before SegmentationGenerator.__getitem__ turns it into float tensors (utils.py:375-402): the
        batch crosses PCIe as bytes on a copy stream while the previous step runs, and X / Y / SW are produced on the device
        (feed.BatchFeeder: widening copy + dl3_prepare_targets).
        Under distribute() (utils.py:209-211 + :231-241): global_batch=True — the generator yields the GLOBAL batch and each
        rank stages only its contiguous shard (rows dp.shard(n)) — or global_batch=False — the generator is already sharded
        by rank (e.g. `idx[rank::world]`) and yields this rank's images only.  Either way only the shard crosses PCIe."""
        self._check_fit_kw(kw)
        hist = []
        steps = steps_per_epoch or len(generator)
        if device_feed:
            return self._fit_device_feed(generator, steps, epochs, n_classes, global_batch)
        for _ in range(epochs):
            for i in range(steps):
                item = generator[i] if hasattr(generator, "__getitem__") else next(generator)
                X, Y = item[0], item[1]
                SW = item[2] if len(item) > 2 else None
                if isinstance(SW, dict):
                    SW = list(SW.values())[0]
                hist.append(self.train_on_batch(X, Y, SW, lazy_loss=True, global_batch=global_batch))
            hist = [float(l) for l in hist]   # one read per epoch
            if hasattr(generator, "on_epoch_end"):
                generator.on_epoch_end()
        return hist

    def _fit_device_feed(self, generator, steps, epochs, n_classes, global_batch=True):
        from .feed import BatchFeeder
        dp = self._dp if self._dp_active() else None
        ekw = dict(external_nnz=True) if dp is not None else {}
        opt = (self._compiled or {}).get("optimizer") or {}
        C = int(n_classes if n_classes is not None else self.output.shape[-1])
        hist, feeders = [], {}
        for _ in range(epochs):
            def batches():
                for i in range(steps):
                    item = generator[i] if hasattr(generator, "__getitem__") else next(generator)
                    X, L = np.asarray(item[0]), np.asarray(item[1])
                    if dp is not None and global_batch:
                        # the rank's contiguous shard of the global batch (the remainder of a batch that does not divide goes
                        # to the last rank, like multi_gpu_model's last tower): only these rows are staged and copied
                        if X.shape[0] < dp.world:
                            raise ValueError("global batch of %d images cannot be split over %d ranks" % (X.shape[0], dp.world))
                        lo, hi = dp.shard(X.shape[0])
                        X, L = X[lo:hi], L[lo:hi]
                    yield X, L
            losses = []
            it = iter(batches())
            first = next(it, None)
            if first is None:
                break
            eng = self._engine(first[0].shape[0], True, **ekw)
            if dp is not None:
                self._dp_sync_weights(eng)
            key = (id(eng), first[1].dtype.str)
            if key not in feeders:
                feeders[key] = BatchFeeder(eng, C, np.uint8 if first[1].dtype == np.uint8 else np.int32)

            def chain():
                yield first
                for b in it:
                    if b[0].shape[0] != eng.B:
                        raise ValueError("device_feed needs batches of one size (got %d after %d)" % (b[0].shape[0], eng.B))
                    yield b

            def step():
                eng.fwd_bwd()
                if dp is not None:
                    # ONE all-reduce of the arena (gradients + the shard's count(w != 0) and loss sum) behind the replayed
                    # graph, the scale finished on the device (Engine.train_step does the same for host-array batches)
                    dp.allreduce_grads(eng.grads)
                eng.adam(opt)   # (norm defaults to the engine's external_nnz)
                losses.append(eng.loss_handle())

            feeders[key].run(chain(), step)
            hist += [float(l) for l in losses]   # one read per epoch
            if hasattr(generator, "on_epoch_end"):
                generator.on_epoch_end()
        return hist
