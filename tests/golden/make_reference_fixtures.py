"""Pins the oracle to the reference ITSELF wherever that is possible without TensorFlow.

Run from the repo root in a container that has /root/reference:   python tests/golden/make_reference_fixtures.py

(i) Always (no Keras/TF needed) -> tests/golden/reference_structure.json
    * `_make_divisible` (deeplabv3p.py:157-164) is EXECUTED: its source is cut out of the reference file with `ast` and
      exec'd as it stands (it is pure Python), then evaluated on a grid of (v, divisor) — inputs and outputs are stored.
    * everything structural that the reference states as literals is EXTRACTED with `ast` (no code is copied, only
      argument values): every `_inverted_res_block(...)` and `_xception_block(...)` call of Deeplabv3() with its keyword
      values, the OS-dependent stride / rate tuples, each BatchNormalization call's epsilon / momentum, every
      `SepConv_BN(...)` call's rate / depth_activation / epsilon, the Deeplabv3() / Subpixel / SegModel signature
      defaults, the string literals used as layer names, do_crf's numeric parameters.
    tests/test_reference_fixtures.py checks BOTH restatements (oracle/dl3_oracle.py, oracle/torch_ref.py) and the
    product's own tables (keras-segmentation-deeplab-v3.1_amd/deeplabv3p.py) against this file.
(ii) Only where `import keras, tensorflow` works (Keras 2.2.4 / TF 1.13, SURVEY §8c — NOT in the build container):
    regenerates tests/golden/cfg1_mnv2_128_c2.npz and tests/golden/ops.npz from the REFERENCE in the very format
    make_golden.py writes from the oracle (same seeds, same keys), plus tests/golden/reference_train_step.npz (loss,
    a few gradients and the BatchNorm moving statistics after one train_on_batch).  This is the hand-off: on such a box
    `python tests/golden/make_reference_fixtures.py --from-reference` turns "parity unpinned" into pinned.
"""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DL3_REFERENCE_DIR", "/root/reference")


def _src(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def _val(node, src):
    """literal value of an AST node, or its source text when it is not a literal"""
    try:
        return ast.literal_eval(node)
    except Exception:
        return {"expr": ast.get_source_segment(src, node)}


def _calls(tree, fname):
    out = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Call) and ((isinstance(n.func, ast.Name) and n.func.id == fname) or
                                        (isinstance(n.func, ast.Attribute) and n.func.attr == fname)):
            out.append(n)
    return sorted(out, key=lambda c: (c.lineno, c.col_offset))


def _funcdef(tree, name):
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == name:
            return n
    raise KeyError(name)


def _defaults(fn, src):
    a = fn.args
    names = [x.arg for x in a.args]
    d = [None] * (len(names) - len(a.defaults)) + list(a.defaults)
    return [[n, (_val(v, src) if v is not None else "<required>")] for n, v in zip(names, d)]


def structure():
    src = _src("deeplabv3p.py")
    tree = ast.parse(src)
    out = {"reference_files": {}, "source": "extracted by tests/golden/make_reference_fixtures.py with ast (values only)"}
    import hashlib
    for f in ("deeplabv3p.py", "subpixel.py", "utils.py"):
        out["reference_files"][f] = hashlib.sha256(_src(f).encode()).hexdigest()

    # ---- (i-a) executed: _make_divisible, cut out of the reference and run as it stands
    fn = _funcdef(tree, "_make_divisible")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, "deeplabv3p.py"), "exec"), ns)
    md = ns["_make_divisible"]
    grid = []
    for divisor in (8,):
        for alpha in (0.35, 0.5, 0.75, 1.0, 1.3, 1.4):
            for filters in (8, 16, 24, 32, 64, 96, 160, 320, 1280):
                grid.append([filters * alpha, divisor, int(md(filters * alpha, divisor))])
                grid.append([int(filters * alpha), divisor, int(md(int(filters * alpha), divisor))])
    for v in range(1, 200):
        grid.append([v, 8, int(md(v, 8))])
    out["make_divisible"] = {"line": fn.lineno, "cases": grid}

    # ---- (i-b) extracted: the MobileNetV2 body
    blocks = []
    for c in _calls(tree, "_inverted_res_block"):
        kw = {k.arg: _val(k.value, src) for k in c.keywords}
        kw.setdefault("rate", 1)  # the def's default (checked below)
        kw["line"] = c.lineno
        blocks.append(kw)
    out["inverted_res_blocks"] = blocks
    out["inverted_res_block_signature"] = _defaults(_funcdef(tree, "_inverted_res_block"), src)
    # ---- the Xception body
    xb = []
    for c in _calls(tree, "_xception_block"):
        pos = [_val(a, src) for a in c.args]
        kw = {k.arg: _val(k.value, src) for k in c.keywords}
        xb.append({"line": c.lineno, "args": pos, "kwargs": kw})
    out["xception_blocks"] = xb
    out["xception_block_signature"] = _defaults(_funcdef(tree, "_xception_block"), src)
    # OS-dependent constants: every tuple/int assignment to these names, in source order
    names = ("entry_block3_stride", "middle_block_rate", "exit_block_rates", "atrous_rates", "OS")
    assigns = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and n.targets[0].id in names:
            assigns.append([n.lineno, n.targets[0].id, _val(n.value, src)])
    out["os_constants"] = sorted(assigns)
    out["middle_flow_repeats"] = [_val(c.args[0], src) for c in _calls(tree, "range")]
    # SepConv_BN call sites and its own definition
    out["sepconv_signature"] = _defaults(_funcdef(tree, "SepConv_BN"), src)
    out["sepconv_calls"] = [{"line": c.lineno, "args": [_val(a, src) for a in c.args],
                             "kwargs": {k.arg: _val(k.value, src) for k in c.keywords}} for c in _calls(tree, "SepConv_BN")]
    out["conv2d_same_calls"] = [{"line": c.lineno, "args": [_val(a, src) for a in c.args],
                                 "kwargs": {k.arg: _val(k.value, src) for k in c.keywords}} for c in _calls(tree, "_conv2d_same")]
    # every BatchNormalization / Conv2D / DepthwiseConv2D / Dropout / AveragePooling2D construction
    for lname in ("BatchNormalization", "Conv2D", "DepthwiseConv2D", "Dropout", "ZeroPadding2D"):
        out[lname + "_calls"] = [{"line": c.lineno, "args": [_val(a, src) for a in c.args],
                                  "kwargs": {k.arg: _val(k.value, src) for k in c.keywords}} for c in _calls(tree, lname)]
    out["Deeplabv3_signature"] = _defaults(_funcdef(tree, "Deeplabv3"), src)
    out["string_literals"] = sorted({n.value for n in ast.walk(tree) if isinstance(n, ast.Constant) and isinstance(n.value, str)
                                     and len(n.value) < 60 and "\n" not in n.value})
    # ---- subpixel.py / utils.py
    ssrc = _src("subpixel.py")
    stree = ast.parse(ssrc)
    sub = _funcdef(stree, "Subpixel")
    out["Subpixel_init_signature"] = _defaults(_funcdef(sub, "__init__"), ssrc)
    out["ICNR_init_signature"] = _defaults(_funcdef(_funcdef(stree, "ICNR"), "__init__"), ssrc)
    out["icnr_weights_signature"] = _defaults(_funcdef(stree, "icnr_weights"), ssrc)
    usrc = _src("utils.py")
    utree = ast.parse(usrc)
    seg = _funcdef(utree, "SegModel")
    out["SegModel_class_attrs"] = {t.targets[0].id: _val(t.value, usrc) for t in seg.body
                                   if isinstance(t, ast.Assign) and isinstance(t.targets[0], ast.Name)}
    out["SegModel_init_signature"] = _defaults(_funcdef(seg, "__init__"), usrc)
    out["create_seg_model_signature"] = _defaults(_funcdef(seg, "create_seg_model"), usrc)
    crf = _funcdef(utree, "do_crf")
    out["do_crf_signature"] = _defaults(crf, usrc)
    out["do_crf_calls"] = [{"fn": (c.func.attr if isinstance(c.func, ast.Attribute) else getattr(c.func, "id", "?")),
                            "line": c.lineno, "args": [_val(a, usrc) for a in c.args],
                            "kwargs": {k.arg: _val(k.value, usrc) for k in c.keywords}}
                           for c in ast.walk(crf) if isinstance(c, ast.Call)]
    out["do_crf_calls"] = sorted(out["do_crf_calls"], key=lambda d: d["line"])
    return out


def from_reference():
    """(ii) regenerate the golden vectors from the reference proper.  Needs Keras 2.2.4 + TF 1.13."""
    import keras  # noqa: F401
    from keras import backend as K
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import deeplabv3p as R  # the reference module
    from oracle import dl3_oracle as O
    from tests.golden.make_golden import cfg1_case

    def load(model, params):
        for l in model.layers:
            ws = l.weights
            if ws:
                l.set_weights([params[w.name] for w in ws])  # Keras weight names == the params keys ("<layer>/kernel:0")

    # cfg1: same seeds, same calibration batch, same keys as make_golden.main()
    kw, params, x, _ = cfg1_case()  # the oracle's calibration decides the moving statistics both sides use
    model = R.Deeplabv3(weights=None, input_shape=(128, 128, 3), classes=2, backbone="mobilenetv2", OS=16, infer=True)
    load(model, params)
    logits_layer = [l for l in model.layers if l.name in ("custom_logits_semantic", "logits_semantic")][0]
    pre = K.function([model.input, K.learning_phase()], [model.layers[-2].input if False else model.layers[-1].input])
    logits = pre([x, 0])[0]
    del logits_layer
    rng = np.random.default_rng(123)
    idx = rng.integers(0, logits.size, 256)
    flat = logits.reshape(-1)
    bn = {k: v for k, v in params.items() if "/moving_" in k}
    np.savez_compressed(
        os.path.join(HERE, "cfg1_mnv2_128_c2.npz"),
        sample_index=idx, sample_logits=flat[idx], logits_sum=np.float64(flat.astype(np.float64).sum()),
        logits_abs_sum=np.float64(np.abs(flat.astype(np.float64)).sum()),
        argmax_sum=np.int64(logits.argmax(-1).sum()), argmax=np.packbits(logits.argmax(-1).astype(np.uint8)),
        logits_max=np.float32(np.abs(flat).max()), **{"bn:" + k: v for k, v in bn.items()})
    # per-op vectors: the Keras layers themselves on the tiny tensors of make_golden.main()
    from keras.layers import DepthwiseConv2D, Input, Lambda, ZeroPadding2D
    from keras.models import Model
    import tensorflow as tf
    rng = np.random.default_rng(7)
    xs = rng.normal(0, 1, (1, 6, 7, 4)).astype(np.float32)
    w = rng.normal(0, 1, (3, 3, 4)).astype(np.float32)
    out = {}
    for s, r in ((1, 1), (1, 2), (2, 1), (1, 5)):
        inp = Input((6, 7, 4))
        y = DepthwiseConv2D(3, strides=s, dilation_rate=r, padding="same", use_bias=False)(inp)
        m = Model(inp, y)
        m.layers[-1].set_weights([w[..., None]])
        out["dw_s%d_r%d" % (s, r)] = m.predict(xs)
    for (ho, wo) in ((13, 20), (48, 56)):
        inp = Input((6, 7, 4))
        m = Model(inp, Lambda(lambda t, sz=(ho, wo): tf.image.resize_bilinear(t, sz))(inp))
        out["resize_6x7_to_%dx%d" % (ho, wo)] = m.predict(xs)
    sys.path.insert(0, REF)
    from subpixel import Subpixel
    I = rng.normal(0, 1, (1, 2, 3, 2 * 9)).astype(np.float32)
    sp = Subpixel(2, 1, 3, padding="same")
    inp = Input((2, 3, 18))
    m = Model(inp, Lambda(lambda t: sp._phase_shift(t))(inp))
    out["phase_shift_r3"] = m.predict(I)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), x=xs, w=w, I=I, **out)
    # one training step of the notebook's compile() on cfg1's model shape: loss, gradients, moving statistics
    sys.path.insert(0, REF)
    import utils as RU
    from keras.optimizers import Adam
    tm = R.Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=3, backbone="mobilenetv2", OS=16)
    p3 = O.init_params(O.param_shapes("mobilenetv2", 3), seed=1)
    load(tm, p3)
    tm.compile(optimizer=Adam(lr=7e-4, epsilon=1e-8, decay=1e-6), sample_weight_mode="temporal",
               loss=RU.sparse_crossentropy_ignoring_last_label)
    r0 = np.random.default_rng(0)
    xb = r0.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    yb = r0.integers(0, 4, (2, 64 * 64, 1)).astype(np.float32)
    sw = ((yb[:, :, 0] < 3) * r0.uniform(0.5, 2.0, (2, 64 * 64))).astype(np.float32)
    loss = tm.train_on_batch(xb, yb, sample_weight=sw)
    after = {w_.name: v for l in tm.layers for w_, v in zip(l.weights, l.get_weights())}
    np.savez_compressed(os.path.join(HERE, "reference_train_step.npz"), loss=np.float64(loss), x=xb, y=yb, sw=sw,
                        **{"after:" + k: v for k, v in after.items()})
    print("golden vectors regenerated from the reference (Keras %s / TF %s)" % (keras.__version__, tf.__version__))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not found at %s (it exists only in the build container)" % REF)
    st = structure()
    path = os.path.join(HERE, "reference_structure.json")
    with open(path, "w") as f:
        json.dump(st, f, indent=1, sort_keys=True)
    print("wrote %s: %d _make_divisible cases, %d inverted-res blocks, %d xception block calls" % (
        path, len(st["make_divisible"]["cases"]), len(st["inverted_res_blocks"]), len(st["xception_blocks"])))
    if "--from-reference" in sys.argv:
        from_reference()
    else:
        try:
            import keras  # noqa: F401
            import tensorflow  # noqa: F401
            print("Keras/TensorFlow are importable here: run with --from-reference to regenerate the golden vectors")
        except ImportError:
            print("Keras/TensorFlow not importable: golden vectors stay oracle-generated (parity unpinned, DESIGN.md §4)")


if __name__ == "__main__":
    main()
