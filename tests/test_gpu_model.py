"""-m gpu parity tests of the whole path: Deeplabv3() / SegModel heads on libdl3.so vs the CPU oracle
on the same seeded inputs and weights.

Tolerances (north star): logits within 1e-3 relative (fp32), argmax masks bit-exact.  Gradients are
compared against the float64 oracle by relative L2 error per tensor — an fp32 run of the ORACLE itself
differs from its float64 run by up to ~1e-2 on single elements after 50 BatchNorm backward passes
(see tests/test_oracle.py::test_fp32_noise_floor), so element-wise 1e-3 would test rounding, not code.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dl3_oracle as O
from tests.gpu_util import dropout_keep_mask, relerr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(backbone="mobilenetv2", input_shape=(64, 64, 3), classes=3, head="deeplab", seed=1):
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    from dl3_amd.utils import SegModel
    G.clear_session()
    if head == "deeplab":
        model = Deeplabv3(weights=None, input_shape=input_shape, classes=classes, backbone=backbone, OS=16)
    else:
        model = SegModel(image_size=input_shape[:2]).create_seg_model(head, n=classes, backbone=backbone)
    shapes = O.param_shapes(backbone, classes, head=head)
    params = O.init_params(shapes, seed=seed)
    return model, params


def _load(model, params):
    names = set()
    for l in model.layers:
        if l.weights:
            l.set_weights([params[n] for n in l.weights])
            names.update(l.weights)
    assert names == set(params), (sorted(names ^ set(params))[:5])


def _l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _assert_argmax_parity(mask, ref_logits, rel=1e-3):
    """masks must be identical except where the ORACLE's own top-2 margin is below the fp32 agreement of the logits
    (`rel`, 1e-3 relative unless the case is documented as ill-conditioned): a different fp32 summation order may
    legitimately flip such near-ties.  The count is reported."""
    ref_mask = ref_logits.argmax(-1)
    diff = mask != ref_mask
    if diff.any():
        srt = np.sort(ref_logits, axis=-1)
        margin = (srt[..., -1] - srt[..., -2])[diff]
        assert margin.max() < rel * np.abs(ref_logits).max(), ("argmax flipped on a clear winner", int(diff.sum()), margin.max())
        assert diff.mean() < 2e-3, int(diff.sum())
        print("argmax near-tie flips: %d of %d pixels (max margin %.2e)" % (diff.sum(), diff.size, margin.max()))


def test_cfg1_inference_matches_golden_and_oracle():
    """BASELINE.json configs[0]: mobilenetv2 128x128, 2 classes, single-image forward."""
    from tests.golden.make_golden import cfg1_case
    kw, params, x, ref_logits = cfg1_case()
    model, _ = _build(input_shape=(128, 128, 3), classes=2)
    _load(model, params)
    probs = model.predict(x, batch_size=1)
    assert probs.shape == (1, 128 * 128, 2)
    eng = model._active
    logits = eng.logits()
    assert relerr(logits, ref_logits) < 1e-3
    assert np.array_equal(eng.argmax(), ref_logits.argmax(-1)), "argmax mask not bit-exact vs oracle"
    assert np.allclose(probs.reshape(ref_logits.shape), O.softmax(ref_logits), atol=1e-4)
    g = np.load(os.path.join(GOLD, "cfg1_mnv2_128_c2.npz"))
    assert np.allclose(logits.reshape(-1)[g["sample_index"]], g["sample_logits"], atol=1e-3 * float(g["logits_max"]))
    assert np.array_equal(np.packbits(eng.argmax().astype(np.uint8)), g["argmax"])


def _train_case(backbone, input_shape, classes, head, B, dropout):
    model, params = _build(backbone, input_shape, classes, head)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (B,) + input_shape).astype(np.float32)
    kw = dict(backbone=backbone, input_shape=input_shape, classes=classes, OS=16, head=head)
    params = O.calibrate_bn(params, x, **kw)
    _load(model, params)
    H, W = input_shape[:2]
    labels = rng.integers(0, classes + 1, (B, H * W)).astype(np.float32)
    sw = ((labels < classes) * rng.uniform(0.5, 2.0, labels.shape)).astype(np.float32)
    eng = model._engine(B, True, dropout=dropout, use_graph=False)
    eng.set_input(x)
    eng.set_targets(labels, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    mask = None
    if dropout:
        fh, fw = H // 8, W // 8
        mask = dropout_keep_mask(eng.seed, B * fh * fw * 256, 0.1).reshape(B, fh, fw, 256).astype(np.float64)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, logits, net = O.train_grads(p64, x.astype(np.float64), labels.astype(np.float64),
                                              sw.astype(np.float64), dropout_mask=mask, **kw)
    return model, eng, loss, grads, logits, net


@pytest.mark.parametrize("head,dropout", [("deeplab", False), ("deeplab", True), ("subpixel", False),
                                          ("original", False)])
def test_mnv2_train_step_gradients(head, dropout):
    classes = 3
    model, eng, loss, grads, logits, net = _train_case("mobilenetv2", (64, 64, 3), classes, head, 2, dropout)
    assert relerr(eng.logits(), logits) < 1e-3
    assert abs(float(eng.loss[0].item()) - loss) < 1e-4 * abs(loss)
    worst, wname = 0.0, None
    num = den = 0.0
    for name, g in grads.items():
        if g is None or "/moving_" in name:
            continue
        gn = np.abs(g).max()
        got = eng.grad_of(name)
        if gn < 1e-6:  # structurally zero gradients (beta in front of a batch-norm'd conv)
            assert np.abs(got).max() < 1e-4, name
            continue
        e = _l2(got, g)
        num += float(np.sum((got.astype(np.float64) - g) ** 2))
        den += float(np.sum(g ** 2))
        if e > worst:
            worst, wname = e, name
    # fp32 noise floor: the ORACLE's own fp32 run differs from its float64 run by 7e-3 (whole gradient vector) and
    # 1.1e-2 (worst tensor) on this case (tests/test_oracle.py::test_fp32_noise_floor); late layers agree to 5e-5
    assert np.sqrt(num / den) < 2e-2, np.sqrt(num / den)
    assert worst < 5e-2, (wname, worst)
    assert _l2(eng.grad_of("concat_projection/kernel:0"), grads["concat_projection/kernel:0"]) < 1e-3
    # BatchNorm moving statistics (TF FusedBatchNorm semantics)
    for name, st in list(net.new_stats.items())[:8]:
        layer = model.get_layer(name)
        mm, mv = layer.get_weights()[2:]
        assert relerr(mm, st["mean"]) < 1e-3 and relerr(mv, st["var"]) < 1e-3, name


@pytest.mark.parametrize("OS", [16, 8])
def test_xception_train_step_gradients(OS):
    """BASELINE.json configs[3] architecture (Xception entry/middle/exit flow, 5-branch ASPP, decoder) at a size the
    float64 oracle finishes in seconds; OS=8 exercises the rate 12/24/36 atrous branches on a small map."""
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    shape, classes, B = (64, 64, 3), 3, 2
    G.clear_session()
    model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone="xception", OS=OS)
    kw = dict(backbone="xception", input_shape=shape, classes=classes, OS=OS)
    params = O.init_params(O.param_shapes("xception", classes), seed=1)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    params = O.calibrate_bn(params, x, **kw)
    _load(model, params)
    # inference path first (the training step below updates the moving statistics): logits 1e-3, argmax exact
    model.predict(x, batch_size=B)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref, _ = O.forward(p64, x.astype(np.float64), **kw)
    ref32, _ = O.forward(params, x, **kw)
    # at OS=16 the deepest maps of a 64x64 input are 4x4 (32 samples per BN channel): ill-conditioned in fp32 —
    # the fp32 ORACLE's own distance to float64 is the yardstick there
    tol = max(1e-3, 2.0 * relerr(ref32, ref))
    assert relerr(model._active.logits(), ref) < tol, (relerr(model._active.logits(), ref), tol)
    _assert_argmax_parity(model._active.argmax(), ref, rel=tol)
    labels = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
    sw = (labels < classes).astype(np.float32)
    eng = model._engine(B, True, dropout=False, use_graph=False)
    eng.set_input(x)
    eng.set_targets(labels, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    loss, grads, logits, net = O.train_grads(p64, x.astype(np.float64), labels.astype(np.float64),
                                              sw.astype(np.float64), **kw)
    assert relerr(eng.logits(), logits) < 1e-3
    assert abs(float(eng.loss[0].item()) - loss) < 1e-4 * abs(loss)
    # yardstick for the deep, ill-conditioned tensors: the fp32 run of the ORACLE itself against its float64 run
    _, grads32, _, _ = O.train_grads(params, x, labels, sw, **kw)
    for name in ("decoder_conv1_pointwise/kernel:0", "aspp3_depthwise/depthwise_kernel:0", "feature_projection0/kernel:0",
                 "exit_flow_block1_shortcut/kernel:0", "middle_flow_unit_8_separable_conv2_pointwise/kernel:0"):
        # (OS=16 on a 64x64 input ends in 4x4 maps, 32 samples per BatchNorm channel: a different fp32 summation order
        # moves these tensors by a few times the oracle's own fp32 rounding distance; OS=8 stays below 2e-2)
        tol_g = max(2e-2, 4.0 * _l2(grads32[name], grads[name]))
        assert _l2(eng.grad_of(name), grads[name]) < tol_g, (name, tol_g)
    num = den = 0.0
    for name, g in grads.items():
        if g is None or "/moving_" in name or np.abs(g).max() < 1e-6:
            continue
        num += float(np.sum((eng.grad_of(name).astype(np.float64) - g) ** 2))
        den += float(np.sum(g ** 2))
    assert np.sqrt(num / den) < 5e-2, np.sqrt(num / den)


def test_graph_replay_and_determinism():
    """hipGraph replay == eager launch sequence, and two runs are bit-identical (no float atomics).  With Dropout on, the
    mask changes every step (device-side step counter), so the replayed graph must walk the same mask sequence as eager
    launches do."""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    labels = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    _load(model, params)
    eng = model._engine(2, True, dropout=False, use_graph=True)
    eng.set_input(x)
    eng.set_targets(labels)
    eng.fwd_bwd()  # eager
    g0 = eng.grads.clone()
    l0 = float(eng.loss[0].item())
    eng.fwd_bwd()  # captured + replayed
    assert eng.graph is not None, "hipGraph capture did not happen"
    g1 = eng.grads.clone()
    eng.fwd_bwd()
    g2 = eng.grads.clone()
    assert torch.equal(g0, g1) and torch.equal(g1, g2)
    assert float(eng.loss[0].item()) == l0
    # dropout on: eager steps 0..3 against (eager step 0 + replayed steps 1..3), same seed
    seq = {}
    for use_graph in (False, True):
        e = model._engine(2, True, dropout=True, use_graph=use_graph)
        e.drop_step.zero_()
        e.set_input(x)
        e.set_targets(labels)
        seq[use_graph] = []
        for _ in range(4):
            e.fwd_bwd()
            seq[use_graph].append(e.grads.clone())
        assert (e.graph is not None) == use_graph
    for a_, b_ in zip(seq[False], seq[True]):
        assert torch.equal(a_, b_)
    assert not torch.equal(seq[True][1], seq[True][2])  # a new mask every step


def test_dropout_mask_changes_every_step():
    """Dropout(0.1) (deeplabv3p.py:410) must draw a new keep mask every training step — also when the step is a replayed
    hipGraph, whose launch arguments are frozen (the step number lives in device memory) — and a different one on every
    data-parallel rank."""
    from dl3_amd.engine import MaterializeUnit
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    labels = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    eng = model._engine(2, True, dropout=True, use_graph=True)
    unit = [u for u in eng.units if isinstance(u, MaterializeUnit)][0]
    eng.set_input(x)
    eng.set_targets(labels)
    masks = []
    for it in range(4):  # step 0 eager, steps 1-3 captured / replayed
        assert int(eng.drop_step.item()) == it
        eng.fwd_bwd()
        torch.cuda.synchronize()
        out = unit.outv.buf.t.cpu().numpy()
        want = dropout_keep_mask(eng.seed, out.size, 0.1, step=it)
        # concat_projection's output is relu'd (>= 0): a kept element may be 0 too, a dropped one is always 0
        assert np.all(out[want == 0] == 0)
        masks.append(out != 0)
        kept_nonzero = (out != 0).sum() / max((want != 0).sum(), 1)
        assert kept_nonzero > 0.2, kept_nonzero
    assert eng.graph is not None
    for a in range(4):
        for b in range(a + 1, 4):
            assert not np.array_equal(masks[a], masks[b])
    # another rank: another stream of masks
    eng1 = model._engine(2, True, dropout=True, use_graph=False, rank=1)
    assert eng1.seed != eng.seed
    assert not np.array_equal(dropout_keep_mask(eng1.seed, 4096, 0.1), dropout_keep_mask(eng.seed, 4096, 0.1))


def test_optimizer_state_survives_a_batch_size_change():
    """a last, smaller batch of an epoch must continue the same Adam (moments, iteration), not start a fresh one"""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(8)
    x = rng.integers(0, 256, (4, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 3, (4, 64 * 64, 1)).astype(np.float32)
    model.compile(optimizer=dict(lr=7e-4))
    model.train_on_batch(x, y, dropout=False)
    model.train_on_batch(x, y, dropout=False)
    e4 = model._active
    m4 = e4.adam_m.clone()
    model.train_on_batch(x[:2], y[:2], dropout=False)
    e2 = model._active
    assert e2 is not e4 and e2.iteration == 3 and e4.iteration == 2
    # the moments were carried over and then updated once: m = 0.9*m4 + 0.1*g
    g = e2.grads[:e2.n_param]   # (behind the gradients: the data-parallel arena tail)
    assert torch.allclose(e2.adam_m, 0.9 * m4 + 0.1 * g, rtol=1e-4, atol=1e-6 * float(g.abs().max()))


def test_training_reduces_loss_and_weights_roundtrip(tmp_path):
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 3, (2, 64 * 64, 1)).astype(np.float32)
    model.compile(optimizer=dict(lr=7e-4, epsilon=1e-8, decay=1e-6))
    losses = [model.train_on_batch(x, y) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    p1 = model.predict(x, batch_size=2)
    path = str(tmp_path / "w.npz")
    model.save_weights(path)
    model2, _ = _build(input_shape=(64, 64, 3), classes=3, seed=9)
    model2.load_weights(path, by_name=True)
    p2 = model2.predict(x, batch_size=2)
    assert np.array_equal(p1, p2)


@pytest.mark.parametrize("head", ["deeplab", "subpixel"])
def test_keras_h5_weights_into_the_engine(tmp_path, head):
    """SURVEY §8f N1 on the GPU box: a Keras-layout .h5 written by the package's own HDF5 subset (h5lite; h5py is not
    installed there) is loaded BY NAME into a fresh Deeplabv3() (deeplabv3p.py:465) and POSITIONALLY into a fresh
    SegModel head (utils.py:206-207); the engine's prediction from the loaded file must match the oracle run on the very
    same arrays."""
    shape, classes = (64, 64, 3), 3
    rng = np.random.default_rng(21)
    x = rng.integers(0, 256, (2,) + shape).astype(np.float32)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head=head)
    src, params = _build("mobilenetv2", shape, classes, head, seed=5)
    params = O.calibrate_bn(params, x, **kw)
    _load(src, params)
    path = str(tmp_path / ("mobilenetv2_%s.h5" % head))
    src.save_weights(path)
    with open(path, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    dst, other = _build("mobilenetv2", shape, classes, head, seed=9)  # different initial weights
    if head == "deeplab":
        dst.load_weights(path, by_name=True)
    else:
        dst.load_weights(path)  # positional: weight-bearing layers zipped in file order
    for l in dst.layers:
        for n, w in zip(l.weights.keys(), l.get_weights()):
            assert np.array_equal(w, params[n]), n
    probs = dst.predict(x, batch_size=2)
    ref, _ = O.forward({k: v.astype(np.float64) for k, v in params.items()}, x.astype(np.float64), **kw)
    assert relerr(dst._active.logits(), ref) < 1e-3
    _assert_argmax_parity(dst._active.argmax(), ref)   # identical up to near-ties of the float64 logits themselves
    # logits agree to 1e-3 of max|logit|: probabilities to a quarter of that absolute logit error
    assert np.allclose(probs.reshape(ref.shape), O.softmax(ref), atol=1e-3 * float(np.abs(ref).max()))


def test_full_size_properties():
    """BASELINE.json configs[1] size (512x512x21, B=2): properties that need no oracle run —
    probabilities sum to one, run-to-run bit-identical logits, and the batch is sharded per image
    (image 0's logits do not depend on image 1 in inference mode)."""
    model, _ = _build(input_shape=(512, 512, 3), classes=21)
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (2, 512, 512, 3)).astype(np.float32)
    p = model.predict(x, batch_size=2)
    assert p.shape == (2, 512 * 512, 21)
    assert np.allclose(p.sum(-1), 1.0, atol=1e-5)
    l1 = model._active.logits().copy()
    x2 = x.copy()
    x2[1] = 255 - x2[1]
    model.predict(x2, batch_size=2)
    l2 = model._active.logits()
    assert np.array_equal(l1[0], l2[0]) and not np.array_equal(l1[1], l2[1])


def test_device_targets_and_evaluate():
    """N2 / N3 around the path: targets prepared on the device (utils.prepare_targets = dl3_prepare_targets) drive the
    same training step as the host-prepared ones, and Model.evaluate's device-counted metrics equal the reference's
    host metric definitions evaluated on the predicted probabilities."""
    from dl3_amd import utils as U
    C = 3
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    raw = rng.choice([0, 1, 2, 255], (2, 64, 64), p=[0.6, 0.25, 0.1, 0.05]).astype(np.uint8)
    Yh, SWh, _ = O.prepare_targets(raw.reshape(2, -1), C)
    Yd, SWd = U.prepare_targets(raw, C)
    assert np.array_equal(Yd.cpu().numpy(), Yh) and np.array_equal(SWd.cpu().numpy(), SWh)
    losses = []
    for tgt in ((Yh, SWh), (Yd, SWd)):
        model, params = _build(input_shape=(64, 64, 3), classes=C)
        _load(model, params)
        model.compile(optimizer=dict(lr=7e-4, epsilon=1e-8, decay=1e-6))
        losses.append([model.train_on_batch(x, *tgt) for _ in range(3)])
    assert losses[0] == losses[1], losses
    loss, jac, acc = model.evaluate(x, Yh, batch_size=2)
    probs = model.predict(x, batch_size=2)
    assert jac == U.Jaccard(Yh, probs) and acc == U.sparse_accuracy_ignoring_last_label(Yh, probs)
    ell = U.sparse_crossentropy_ignoring_last_label(Yh, probs)
    # no sample weights: Keras' plain mean over all pixels (void rows have an all-zero one-hot row: ell == 0 there)
    assert np.all(ell[Yh[:, :, 0] == C] == 0)
    assert abs(loss - ell.sum() / ell.size) < 1e-12
    assert 0.0 <= jac <= 1.0 and 0.0 <= acc <= 1.0


def test_frozen_bn_train_step_gradients():
    """bn_mode='frozen' (the notebook's fine-tuning intent): BatchNorm uses the moving statistics in the
    training step, its gamma/beta still receive gradients, the moving statistics stay untouched."""
    classes, B, shape = 3, 2, (64, 64, 3)
    model, params0 = _build("mobilenetv2", shape, classes, "deeplab")
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    # The bar below (5e-3 on the whole gradient vector) holds for a WELL-CONDITIONED case only: the ASPP image-pooling
    # activation (deeplabv3p.py:375-381) is ONE value per image and channel whose ReLU mask gates a term of every pixel's
    # gradient, so a pre-activation within fp32 noise of zero makes any fp32 evaluation a coin flip worth ~1e-2 of the
    # whole vector (profiles/r04_frozen_gradient_noise.txt; this test tripped over exactly that when round 5 changed the
    # branch's summation order).  The batch is therefore chosen, deterministically, as the first of a few seeds whose
    # float64 pre-activations all stay clear of zero.
    from oracle import torch_ref as T
    for seed in (5, 15, 25, 35, 45):
        rng = np.random.default_rng(seed)
        x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
        params = O.calibrate_bn(params0, x, **kw)
        ref = T.Ref(params, True, None, True, torch.float64)
        ref.record = {}
        with torch.no_grad():
            ref.logits(x, **{k: v for k, v in kw.items()})
        z = ref.record["image_pooling_BN"]
        if np.abs(z).min() > 2e-3 * np.abs(z).max():
            break
    else:
        raise AssertionError("no well-conditioned batch among the seeds")
    print("frozen-BN gradient test: seed %d, min |image_pooling pre-activation| %.2e of max %.2e" % (
        seed, float(np.abs(z).min()), float(np.abs(z).max())))
    _load(model, params)
    labels = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
    sw = (labels < classes).astype(np.float32)
    eng = model._engine(B, True, bn_mode="frozen", dropout=False, use_graph=False)
    eng.set_input(x)
    eng.set_targets(labels, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    loss, grads, logits, net = O.train_grads(p64, x.astype(np.float64), labels.astype(np.float64), sw.astype(np.float64),
                                              bn_frozen=True, **kw)
    assert relerr(eng.logits(), logits) < 1e-3
    assert abs(float(eng.loss[0].item()) - loss) < 1e-4 * abs(loss)
    _, g32, _, _ = O.train_grads(params, x, labels, sw, bn_frozen=True, **kw)
    num = den = n32 = 0.0
    per = []
    for name, g in grads.items():
        if g is None or "/moving_" in name or np.abs(g).max() < 1e-6:
            continue
        got = eng.grad_of(name).astype(np.float64)
        num += float(np.sum((got - g) ** 2))
        n32 += float(np.sum((np.asarray(g32[name], np.float64).reshape(g.shape) - g) ** 2))
        den += float(np.sum(g ** 2))
        per.append((float(np.sum((got - g) ** 2)), name, _l2(got, g), _l2(g32[name], g)))
    whole, whole32 = np.sqrt(num / den), np.sqrt(n32 / den)
    print("frozen-BN gradients: whole vector rel-L2 gpu %.2e, numpy-fp32 oracle %.2e; largest contributions:" % (whole, whole32))
    for e2, name, el, e32 in sorted(per, reverse=True)[:6]:
        print("   %-44s share %.2f  rel-L2 gpu %.2e  numpy-fp32 %.2e" % (name, e2 / num, el, e32))
    # frozen BN is a per-channel affine map: the fp32 path is well conditioned
    # (no batch statistics in the chain, far below the 2e-2 floor of the batch-statistics case)
    assert whole < max(5e-3, 2.0 * whole32), (whole, whole32)
    # per tensor: against the fp32 run of the oracle itself (aspp0/kernel: 7e-3 from cancellation in fp32)
    for name in ("Conv_BN/gamma:0", "expanded_conv_16_project_BN/beta:0", "aspp0/kernel:0", "Conv/kernel:0"):
        assert _l2(eng.grad_of(name), grads[name]) < max(5e-3, 2.0 * _l2(g32[name], grads[name])), name
    eng.sync_all_to_host()
    mm = model.get_layer("Conv_BN").get_weights()[2]
    assert np.array_equal(mm, params["Conv_BN/moving_mean:0"])


def test_single_image_batch_statistics_trap():
    """with B=1 the image-pooling BatchNorm sees one sample per channel (variance 0): its output is beta and
    no gradient reaches image_pooling.  The step must stay finite and reproduce exactly that."""
    classes, shape = 3, (64, 64, 3)
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    _load(model, params)
    rng = np.random.default_rng(6)
    x = rng.integers(0, 256, (1,) + shape).astype(np.float32)
    labels = rng.integers(0, classes + 1, (1, shape[0] * shape[1])).astype(np.float32)
    eng = model._engine(1, True, dropout=False, use_graph=False)
    eng.set_input(x)
    eng.set_targets(labels)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    assert np.isfinite(float(eng.loss[0].item()))
    assert np.isfinite(eng.grads.cpu().numpy()).all()
    assert np.abs(eng.grad_of("image_pooling/kernel:0")).max() < 1e-6
    assert np.abs(eng.grad_of("aspp0/kernel:0")).max() > 1e-6
    # ... and with the epsilon tf.nn.fused_batch_norm actually uses (engine.fused_bn_epsilon: never below 1.001e-5): at
    # variance 0 the layer's 1 / sigma is 1 / sqrt(1.001e-5) and its output is beta exactly
    from dl3_amd.engine import V_INVSTD, V_SCALE, V_SHIFT, V_MEAN
    bufs = [(b, off, C) for b in eng.bufs for bn, off, C in b.bns if bn.name == "image_pooling_BN"]
    assert len(bufs) == 1
    b, off, C = bufs[0]
    inv = b.vec[V_INVSTD, off:off + C].cpu().numpy()
    want = np.float32(1.0 / np.sqrt(np.float64(np.float32(1.001e-5))))
    assert np.allclose(inv, want, rtol=2e-6), (inv[:4], want)
    assert abs(float(want) * np.sqrt(1e-5) - 1.0) > 4e-4   # (1 / sqrt(1e-5) would be 5e-4 away)
    y = b.t.view(b.M, b.ld)[:, off:off + C].cpu().numpy()
    out = y * b.vec[V_SCALE, off:off + C].cpu().numpy() + b.vec[V_SHIFT, off:off + C].cpu().numpy()
    beta = model.get_layer("image_pooling_BN").weights["image_pooling_BN/beta:0"]
    assert np.allclose(out, beta[None, :], atol=1e-3 * max(1.0, float(np.abs(beta).max())))


def test_inference_graph_replay_matches_eager():
    """Model.predict replays the forward launch sequence as a hipGraph from the second call on: same bits as the
    eager first call, also after the input and the weights changed."""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(8)
    x1 = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    x2 = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    p_eager = model.predict(x1, batch_size=2)          # call 1: eager
    p_cap = model.predict(x1, batch_size=2)            # call 2: capture + replay
    p_rep = model.predict(x1, batch_size=2)            # call 3: replay
    assert np.array_equal(p_eager, p_cap) and np.array_equal(p_eager, p_rep)
    q = model.predict(x2, batch_size=2)
    assert not np.array_equal(q, p_eager)
    eng = model._active
    assert eng.graph is not None
    eager = model._engine(2, False, use_graph=False)
    assert np.array_equal(eager.predict(x2), q)
    # new weights reach the replayed graph (device buffers are updated in place)
    layer = model.get_layer("logits_semantic" if False else "custom_logits_semantic")
    w = layer.get_weights()
    layer.set_weights([w[0] * 0.5, w[1] + 0.25])
    q2 = model.predict(x2, batch_size=2)
    assert not np.array_equal(q2, q)
    assert np.array_equal(model._engine(2, False, use_graph=False).predict(x2), q2)


def test_fit_generator_with_device_targets():
    """SegmentationGenerator tensor contract (utils.py:257-409) over in-memory arrays: labels prepared on the device,
    fit_generator consumes (X, Y, {'pred_mask': SW}); the batch content equals the oracle's label preparation."""
    from dl3_amd import utils as U
    C = 3
    rng = np.random.default_rng(21)
    imgs = rng.integers(0, 256, (6, 64, 64, 3)).astype(np.uint8)
    labs = rng.choice([0, 1, 2, 255], (6, 64, 64), p=[0.5, 0.3, 0.15, 0.05]).astype(np.uint8)
    gen = U.SegmentationGenerator(imgs, labs, n_classes=C, batch_size=2, shuffle=False)
    assert len(gen) == 3
    X, Y, sw = gen[1]
    Yr, SWr, _ = O.prepare_targets(labs[2:4].reshape(2, -1), C)
    assert X.dtype == np.float32 and X.shape == (2, 64, 64, 3) and np.array_equal(X, imgs[2:4])
    assert np.array_equal(Y.cpu().numpy(), Yr) and np.array_equal(sw["pred_mask"].cpu().numpy(), SWr)
    with pytest.raises(IndexError):
        gen[3]
    model, params = _build(input_shape=(64, 64, 3), classes=C)
    _load(model, params)
    model.compile(optimizer=dict(lr=7e-4, epsilon=1e-8, decay=1e-6))
    hist = model.fit_generator(gen, epochs=5)
    # (three different batches per epoch, a fresh dropout mask every step: compare epoch means, not single steps)
    assert len(hist) == 15 and all(np.isfinite(hist)) and np.mean(hist[-3:]) < np.mean(hist[:3]), hist
    masks = model.predict_mask(imgs[:4], batch_size=2)
    assert masks.shape == (4, 64, 64) and masks.dtype == np.int32
    assert np.array_equal(masks.reshape(4, -1), model.predict(imgs[:4], batch_size=2).argmax(-1))
    # uint8 images cross PCIe as bytes and are widened on the device: same result as the float32 array
    assert np.array_equal(model.predict(imgs[:2], batch_size=2), model.predict(imgs[:2].astype(np.float32), batch_size=2))
    # same losses as feeding the host-prepared tensors batch by batch
    model2, _ = _build(input_shape=(64, 64, 3), classes=C)
    _load(model2, params)
    model2.compile(optimizer=dict(lr=7e-4, epsilon=1e-8, decay=1e-6))
    ref = []
    for _ in range(5):
        for i in range(3):
            Yh, SWh, _ = O.prepare_targets(labs[2 * i:2 * i + 2].reshape(2, -1), C)
            ref.append(model2.train_on_batch(imgs[2 * i:2 * i + 2].astype(np.float32), Yh, SWh))
    assert hist == ref


@pytest.mark.parametrize("shape", [(65, 49, 3), (72, 56, 3)])
def test_odd_and_non_square_inputs(shape):
    """input sizes that are not multiples of the output stride (DeepLab's classic 513 = 8k+1) and non-square maps:
    SAME padding with an odd extent, ceil-mode feature sizes, legacy bilinear between arbitrary sizes"""
    classes, B = 3, 2
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    rng = np.random.default_rng(shape[0])
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    params = O.calibrate_bn(params, x, **kw)
    _load(model, params)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref, _ = O.forward(p64, x.astype(np.float64), **kw)
    probs = model.predict(x, batch_size=B)
    assert probs.shape == (B, shape[0] * shape[1], classes)
    assert relerr(model._active.logits(), ref) < 1e-3
    _assert_argmax_parity(model._active.argmax(), ref)
    labels = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
    sw = (labels < classes).astype(np.float32)
    eng = model._engine(B, True, dropout=False, use_graph=False)
    eng.set_input(x)
    eng.set_targets(labels, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    loss, grads, logits, _ = O.train_grads(p64, x.astype(np.float64), labels.astype(np.float64), sw.astype(np.float64), **kw)
    assert relerr(eng.logits(), logits) < 1e-3
    assert abs(float(eng.loss[0].item()) - loss) < 1e-4 * abs(loss)
    # A 9x7 ASPP map and a batch of two: 126 samples per BatchNorm channel, two for image_pooling_BN.  The logits layer's
    # gradient is well conditioned (1e-5); everything from concat_projection_BN's backward down passes the ReLU masks of
    # the 2 x 256 image-pooling activations and of 126-row maps, and ONE mask that fp32 decides differently moves all of it
    # by ~1e-2.  Round 5, GPU call 13, this very case under four builds that differ in the rounding of the stem
    # convolution / the depthwise kernels: concat_projection/kernel 8.2e-5, 1.0e-4, 1.45e-2, 1.45e-2 — with the logits
    # (1.7e-5) and the loss (1e-6) the same in all four.  The numpy-fp32 run of the oracle (printed) is one more draw.
    _, grads32, _, _ = O.train_grads(params, x, labels, sw, **kw)
    for name, bar in (("custom_logits_semantic/kernel:0", 1e-3), ("concat_projection/kernel:0", 4e-2), ("aspp0/kernel:0", 4e-2)):
        d, yard = _l2(eng.grad_of(name), grads[name]), _l2(grads32[name], grads[name])
        print("   %-36s gradient rel-L2 gpu %.2e, numpy-fp32 oracle %.2e" % (name, d, yard))
        assert d < bar, (name, d, yard)


def test_width_multiplier_alpha():
    """Deeplabv3(alpha=0.5): `_make_divisible` channel rounding (deeplabv3p.py:157-170) through the engine"""
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    shape, classes, B, alpha = (64, 64, 3), 3, 2, 0.5
    G.clear_session()
    model = Deeplabv3(weights=None, input_shape=shape, classes=classes, backbone="mobilenetv2", alpha=alpha)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, alpha=alpha)
    params = O.init_params(O.param_shapes("mobilenetv2", classes, alpha=alpha), seed=3)
    rng = np.random.default_rng(12)
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    params = O.calibrate_bn(params, x, **kw)
    _load(model, params)
    assert model.get_layer("expanded_conv_project").get_weights()[0].shape[-1] == 8
    ref, _ = O.forward({k: v.astype(np.float64) for k, v in params.items()}, x.astype(np.float64), **kw)
    model.predict(x, batch_size=B)
    assert relerr(model._active.logits(), ref) < 1e-3
    _assert_argmax_parity(model._active.argmax(), ref)
    y = rng.integers(0, classes + 1, (B, shape[0] * shape[1], 1)).astype(np.float32)
    l0 = model.train_on_batch(x, y)
    assert np.isfinite(l0)


def test_learns_a_toy_segmentation_task():
    """the whole stack as a user drives it (Deeplabv3 -> compile -> fit_generator on SegmentationGenerator batches ->
    evaluate): bright squares on a dark, noisy background; a few dozen Adam steps must lift Jaccard and accuracy well
    above the untrained model's"""
    from dl3_amd import utils as U
    rng = np.random.default_rng(33)
    n, H, W = 16, 64, 64
    imgs = rng.integers(0, 60, (n, H, W, 3)).astype(np.uint8)
    labs = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        y0, x0 = rng.integers(4, 32, 2)
        s = rng.integers(16, 28)
        imgs[i, y0:y0 + s, x0:x0 + s] = rng.integers(180, 256, 3)
        labs[i, y0:y0 + s, x0:x0 + s] = 1
        labs[i, :2] = 255  # a void border
    model, params = _build(input_shape=(H, W, 3), classes=2)
    _load(model, params)
    model.compile(optimizer=dict(lr=3e-3, epsilon=1e-8, decay=1e-6))
    Yall, SWall, _ = O.prepare_targets(labs.reshape(n, -1), 2)

    def batch_stat_metrics():
        # metrics of the TRAINING-mode forward (batch statistics): with the reference's BatchNorm momentum of 0.999 the
        # moving statistics that predict()/evaluate() use have barely moved after a few dozen steps
        eng = model._engine(8, True)
        eng.set_input(imgs[:8])
        eng.set_targets(Yall[:8], SWall[:8])
        eng.forward()
        probs = O.softmax(eng.logits().astype(np.float64)).reshape(8, -1, 2)
        return U.Jaccard(Yall[:8], probs), U.sparse_accuracy_ignoring_last_label(Yall[:8], probs)

    before = batch_stat_metrics()
    gen = U.SegmentationGenerator(imgs, labs, n_classes=2, batch_size=8, seed=1)
    hist = model.fit_generator(gen, epochs=20)
    after = batch_stat_metrics()
    assert np.isfinite(hist).all() and np.mean(hist[-4:]) < 0.5 * np.mean(hist[:4]), (hist[:4], hist[-4:])
    assert after[0] > max(0.7, before[0] + 0.2), (before, after)    # Jaccard
    assert after[1] > 0.9, (before, after)                          # pixel accuracy
    loss, jac, acc = model.evaluate(imgs.astype(np.float32), Yall, batch_size=8)   # inference path stays consistent
    assert np.isfinite(loss) and 0.0 <= jac <= 1.0 and 0.0 <= acc <= 1.0


@pytest.mark.parametrize("k,padding", [(3, "same"), (2, "valid"), (3, "valid")])
def test_subpixel_with_kernel_size_above_one(k, padding):
    """Subpixel IS a Conv2D with any kernel_size (subpixel.py:42-58; icnr_weights' default shape is 3x3, :9): round 2
    silently lowered kernel_size != 1 as a 1x1 GEMM.  Forward, loss and every gradient of a small graph
    conv3x3/s2 -> BN -> relu -> Subpixel(3, k, r=2) -> softmax against a float64 torch-autograd restatement written
    here (TF SAME padding, the reference's own phase shift, Keras weighted cross-entropy)."""
    import torch.nn.functional as F
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.subpixel import Subpixel, icnr_weights
    G.clear_session(seed=5)
    H, W, B, r, classes, mid = 16, 24, 2, 2, 3, 8
    inp = G.Input(shape=(H, W, 3))
    x = G.Conv2D(mid, 3, strides=2, padding="same", use_bias=False, name="c0")(inp)
    x = G.BatchNormalization(name="c0_BN", epsilon=1e-3)(x)
    x = G.Activation("relu")(x)
    x = Subpixel(classes, k, r, padding=padding, name="sp")(x)
    Hs, Ws = x.shape[0], x.shape[1]
    x = G.Reshape((Hs * Ws, -1))(x)
    x = G.Activation("softmax", name="pred_mask")(x)
    model = G.Model(inp, x, name="sp_test")
    sp = [l for l in model.layers if l.name == "sp"][0]
    kern, bias = sp.get_weights()
    assert kern.shape == (k, k, mid, classes * r * r)
    rng = np.random.default_rng(11)
    sp.set_weights([icnr_weights(scale=r, shape=kern.shape), rng.normal(0, 0.1, bias.shape).astype(np.float32)])
    xin = rng.normal(0, 1, (B, H, W, 3)).astype(np.float32)
    labels = rng.integers(0, classes + 1, (B, Hs * Ws)).astype(np.float32)
    sw = ((labels < classes) * rng.uniform(0.5, 2.0, labels.shape)).astype(np.float32)
    eng = model._engine(B, True, dropout=False, use_graph=False)
    eng.set_input(xin)
    eng.set_targets(labels, sw)
    eng.fwd_bwd()
    torch.cuda.synchronize()
    # ---- float64 restatement
    P = {n: torch.tensor(np.asarray(w, np.float64), requires_grad=True) for l in model.layers for n, w in l.weights.items()
         if "/moving_" not in n}
    xt = torch.tensor(xin.astype(np.float64)).permute(0, 3, 1, 2)
    a = F.conv2d(F.pad(xt, (0, 1, 0, 1)), P["c0/kernel:0"].permute(3, 2, 0, 1), stride=2)  # SAME, even size: pad at the end
    mu, var = a.mean((0, 2, 3), keepdim=True), a.var((0, 2, 3), unbiased=False, keepdim=True)
    a = (a - mu) / torch.sqrt(var + 1e-3) * P["c0_BN/gamma:0"].view(1, -1, 1, 1) + P["c0_BN/beta:0"].view(1, -1, 1, 1)
    a = torch.relu(a)
    if padding == "same":
        a = F.pad(a, ((k - 1) // 2, k - 1 - (k - 1) // 2) * 2)
    y = F.conv2d(a, P["sp/kernel:0"].permute(3, 2, 0, 1)) + P["sp/bias:0"].view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)                                             # [N, a, b, C*r*r]
    Nn, Ha, Wb, _ = y.shape
    # out[n, ia*r+q, ib*r+p, ch] = I[n, ia, ib, ch*r*r + p*r + q]  (subpixel.py:77-88)
    y = y.reshape(Nn, Ha, Wb, classes, r, r).permute(0, 1, 5, 2, 4, 3).reshape(Nn, Ha * r, Wb * r, classes)
    assert (Ha * r, Wb * r) == (Hs, Ws)
    logits = y.reshape(Nn, Hs * Ws, classes)
    t = torch.tensor(labels.astype(np.int64))
    w = torch.tensor(sw.astype(np.float64))
    p = torch.softmax(logits, -1)
    pt = torch.gather(p, 2, t.clamp(max=classes - 1).unsqueeze(-1)).squeeze(-1).clamp(1e-7, 1 - 1e-7)
    ell = torch.where(t < classes, -torch.log(pt), torch.zeros_like(pt))
    loss = (ell * w).sum() / (w != 0).sum()
    loss.backward()
    got = eng.logits().reshape(Nn, Hs * Ws, classes)
    assert relerr(got, logits.detach().numpy()) < 1e-4
    assert abs(float(eng.loss[0].item()) - float(loss)) < 1e-5 * abs(float(loss))
    for n, pt_ in P.items():
        e = _l2(eng.grad_of(n), pt_.grad.numpy())
        print("   %-16s grad rel-L2 %.2e" % (n, e))
        assert e < 2e-4, (n, e)


def test_compile_accepts_the_references_adam_object():
    """segmentation.ipynb cell 2: compile(optimizer=Adam(lr=7e-4, epsilon=1e-8, decay=1e-6), ...) — the object and the
    equivalent dict must train identically (bit for bit: same launches, same scalars)."""
    from dl3_amd.optimizers import Adam
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 3, (2, 64 * 64, 1)).astype(np.float32)
    outs = []
    for opt in (Adam(lr=7e-4, epsilon=1e-8, decay=1e-6), dict(lr=7e-4, epsilon=1e-8, decay=1e-6)):
        model, params = _build(input_shape=(64, 64, 3), classes=3)
        _load(model, params)
        model.compile(optimizer=opt, sample_weight_mode="temporal", loss="sparse_crossentropy_ignoring_last_label")
        losses = [model.train_on_batch(x, y, dropout=False) for _ in range(3)]
        outs.append((losses, model._active.params.cpu().numpy().copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][0][-1] < outs[0][0][0]


def test_recompile_with_the_same_adam_object_keeps_its_iterations():
    """Keras 2.2.4: `optimizer.iterations` belongs to the optimizer OBJECT — compiling again with the same Adam instance (the
    notebook's fine-tuning recompile) re-creates the m / v slots but keeps counting, so lr decay and the bias-correction
    exponent continue; a new object starts at 0."""
    from dl3_amd.optimizers import Adam
    rng = np.random.default_rng(14)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 3, (2, 64 * 64, 1)).astype(np.float32)
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    opt = Adam(lr=7e-4, epsilon=1e-8, decay=1e-2)
    model.compile(optimizer=opt)
    for _ in range(3):
        model.train_on_batch(x, y, dropout=False)
    e1 = model._active
    assert e1.iteration == 3 and float(e1.adam_m.abs().max()) > 0
    model.compile(optimizer=opt)               # same object: the clock goes on, the moments start over
    model.train_on_batch(x, y, dropout=False)
    e2 = model._active
    assert e2 is not e1 and e2.iteration == 4
    # the first step of the new train function is Adam's step t = 4 from zero moments: m = (1 - beta_1) g exactly
    g = e2.grads[:e2.n_param]
    assert torch.allclose(e2.adam_m[:e2.n_param], (1.0 - 0.9) * g, rtol=1e-6, atol=0)
    model.compile(optimizer=Adam(lr=7e-4, epsilon=1e-8, decay=1e-2))   # a NEW object starts at 0
    model.train_on_batch(x, y, dropout=False)
    assert model._active.iteration == 1


def test_backward_fork_is_bit_identical(monkeypatch):
    """DL3_FORK=1 (engine.Engine.run_ops_forked: the 1x1 weight gradients on a second stream / parallel hipGraph branch,
    round 3 — measured slower and off by default) computes exactly what the single-stream plan computes, eagerly and as
    a captured graph."""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(12)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    sw = (y < 3).astype(np.float32)
    got = {}
    # (with the fork on, the weight-gradient launch does not write dY for the bwd-data launch — the two must stay ordered
    # on one stream —, and a materialised dY is rounded once more than the two-tensor operand: same operand form everywhere)
    monkeypatch.setenv("DL3_DY_MAT", "0")
    for fork in ("0", "1", "2"):   # 2 = only the weight gradients paired with a depthwise backward launch leave the chain
        monkeypatch.setenv("DL3_FORK", fork)
        for use_graph in (False, True):
            eng = model._engine(2, True, dropout=False, use_graph=use_graph, seed=100 + int(fork))  # distinct engine keys
            assert eng.fork == (fork != "0") and bool(eng._side) == (fork != "0")
            # the batched weight-gradient fold at the end of the pass waits for the side stream in every fork mode;
            # mode 2 adds a join in front of the chain's next GEMM behind every pair
            assert len(eng._join_before) == (0 if fork == "0" else 1) or (fork == "2" and len(eng._join_before) > 1)
            eng.set_input(x)
            eng.set_targets(y, sw)
            for _ in range(3 if use_graph else 1):   # the graph is captured on the second call
                eng.fwd_bwd()
            torch.cuda.synchronize()
            assert (eng.graph is not None) == use_graph
            got[(fork, use_graph)] = (eng.grads.cpu().numpy().copy(), float(eng.loss[0].item()))
    ref = got[("0", False)]
    for k, v in got.items():
        assert v[1] == ref[1] and np.array_equal(v[0], ref[0]), k


def test_fused_both_gradient_kernel_inside_a_small_plan(monkeypatch):
    """dl3_pwconv_bwd_fused serves layers of >= DL3_FUSED_ROWS pixel rows (32768: nothing in a 64x64 test model).  With the
    threshold at 1 the early 1x1 convolutions of the same model go through it — with a residual addend, with BatchNorm
    sums against another tensor, behind the stem — and the step must agree with the unfused plan to fp32 rounding."""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(13)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    sw = (y < 3).astype(np.float32)
    got = {}
    for i, (rows, on) in enumerate([("1", "1"), ("1", "0")]):
        monkeypatch.setenv("DL3_FUSED_ROWS", rows)
        monkeypatch.setenv("DL3_FUSED_BWD", on)
        eng = model._engine(2, True, dropout=False, use_graph=False, seed=300 + i)
        n = sum(1 for rec in eng.ops_bwd if rec[0] == "dl3_pwconv_bwd_fused")
        assert (n >= 5) == (on == "1"), n
        eng.set_input(x)
        eng.set_targets(y, sw)
        eng.fwd_bwd()
        torch.cuda.synchronize()
        got[on] = (eng.grads.cpu().numpy().copy(), float(eng.loss[0].item()))
    assert got["1"][1] == got["0"][1]
    assert _l2(got["1"][0], got["0"][0]) < 2e-5, _l2(got["1"][0], got["0"][0])


@pytest.mark.parametrize("backbone,shape", [("mobilenetv2", (96, 96, 3)), ("xception", (64, 64, 3))])
def test_poisoned_scratch_changes_nothing(monkeypatch, backbone, shape):
    """DL3_POISON_SCRATCH=1 fills every scratch allocation of the engine (statistic / weight-gradient partial buffers,
    workspaces) with NaN before the plan is built: the kernels own every word they later read — in particular the rows of
    a partial buffer beyond the launch's own grid, which the kernels zero themselves (round 3: no memset nodes) — so
    loss and gradients do not change by a bit."""
    model, params = _build(input_shape=shape, classes=3, backbone=backbone)
    _load(model, params)
    rng = np.random.default_rng(21)
    B = 3   # odd: ragged row tiles and partial counts that differ between the forward and backward decompositions
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    y = rng.integers(0, 4, (B, shape[0] * shape[1])).astype(np.float32)
    sw = (y < 3).astype(np.float32)
    got = {}
    for poison in ("0", "1"):
        monkeypatch.setenv("DL3_POISON_SCRATCH", poison)
        for use_graph in (False, True):
            eng = model._engine(B, True, dropout=False, use_graph=use_graph, seed=300 + int(poison))
            assert eng.poison == (poison == "1")
            eng.set_input(x)
            eng.set_targets(y, sw)
            for _ in range(3 if use_graph else 1):
                eng.fwd_bwd()
            torch.cuda.synchronize()
            got[(poison, use_graph)] = (eng.grads.cpu().numpy().copy(), float(eng.loss[0].item()))
    ref = got[("0", False)]
    assert np.isfinite(ref[1]) and np.isfinite(ref[0]).all()
    for k, v in got.items():
        assert v[1] == ref[1] and np.array_equal(v[0], ref[0]), k


def test_batched_weight_gradient_folds_are_bit_identical(monkeypatch):
    """DL3_BATCH_FOLDS=1 (default: every weight-gradient slab fold of the backward pass in ONE dl3_reduce_partials_batched
    launch at its end, each 1x1 weight gradient keeping its slabs in a workspace of its own) against =0 (a
    dl3_reduce_partials behind every weight-gradient launch): same gradients, bit for bit; ~55 kernel launches less per step."""
    model, params = _build(input_shape=(64, 64, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(22)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    sw = (y < 3).astype(np.float32)
    got, nops = {}, {}
    for batch in ("0", "1"):
        monkeypatch.setenv("DL3_BATCH_FOLDS", batch)
        for use_graph in (False, True):
            eng = model._engine(2, True, dropout=False, use_graph=use_graph, seed=400 + int(batch))
            names = [o[0] for o in eng.ops_bwd]
            assert names.count("dl3_reduce_partials_batched") == int(batch)
            assert (names[-1] == "dl3_reduce_partials_batched") == (batch == "1")
            nops[batch] = len(names)
            eng.set_input(x)
            eng.set_targets(y, sw)
            for _ in range(3 if use_graph else 1):
                eng.fwd_bwd()
            torch.cuda.synchronize()
            got[(batch, use_graph)] = (eng.grads.cpu().numpy().copy(), float(eng.loss[0].item()))
    assert nops["0"] - nops["1"] >= 15   # depthwise / dense-3x3 folds leave the plan (the 1x1 folds were inside their launch's C call)
    ref = got[("0", False)]
    for k, v in got.items():
        assert v[1] == ref[1] and np.array_equal(v[0], ref[0]), k


def test_split_math_under_hipgraph_capture():
    """DL3_GEMM_MATH=split through a captured hipGraph (bench.py's split_math leg): the packed-weight scratch grown by the
    eager first step must serve the capture stream too (round 3: a per-stream scratch made the capture fail); the
    replayed graph reproduces the eager split-math step bit for bit and stays within fp32 round-off of the f32 MFMA."""
    from dl3_amd import capi
    model, params = _build(input_shape=(128, 128, 3), classes=3)
    _load(model, params)
    rng = np.random.default_rng(13)
    B = 16   # 128-row tiles (the split instantiations) need >= 512 row tiles on the early layers
    x = rng.integers(0, 256, (B, 128, 128, 3)).astype(np.float32)
    y = rng.integers(0, 4, (B, 128 * 128)).astype(np.float32)
    sw = (y < 3).astype(np.float32)
    out = {}
    for math in ("f32", "split"):
        capi.set_gemm_math(None if math == "f32" else "split")
        try:
            for use_graph in (False, True):
                eng = model._engine(B, True, dropout=False, use_graph=use_graph, seed=200 + (math == "split"))
                eng.set_input(x)
                eng.set_targets(y, sw)
                for _ in range(3 if use_graph else 1):
                    eng.fwd_bwd()
                torch.cuda.synchronize()
                assert (eng.graph is not None) == use_graph, "hipGraph capture failed in %s math" % math
                out[(math, use_graph)] = eng.grads.cpu().numpy().copy()
        finally:
            capi.set_gemm_math(None)
    assert np.array_equal(out[("split", True)], out[("split", False)])
    assert np.array_equal(out[("f32", True)], out[("f32", False)])
    assert not np.array_equal(out[("split", False)], out[("f32", False)])   # it really is another arithmetic
    assert _l2(out[("split", False)], out[("f32", False)]) < 2e-2           # ... within what separates two fp32 orders


def test_subpixel_head_fused_loss_equals_unfused(monkeypatch):
    """SegModel 'subpixel' head in training: dl3_shuffle_softmax_xent (loss on the unshuffled Subpixel output, no phase
    shift in either direction) against the unfused plan (DL3_FUSE_SHUFFLE=0: phase shift, softmax_xent, inverse shift):
    same loss, same gradients (same arithmetic per pixel; only the order of the loss partial sums differs)"""
    model, params = _build(input_shape=(64, 64, 3), classes=3, head="subpixel")
    _load(model, params)
    rng = np.random.default_rng(21)
    x = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.float32)
    y = rng.integers(0, 4, (2, 64 * 64)).astype(np.float32)
    sw = ((y < 3) * rng.uniform(0.5, 2.0, y.shape)).astype(np.float32)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("DL3_FUSE_SHUFFLE", fuse)
        eng = model._engine(2, True, dropout=False, use_graph=False, seed=300 + int(fuse))
        assert (eng.fused_shuffle is not None) == (fuse == "1")
        names = [o[0] for o in eng.ops_fwd + eng.ops_bwd]
        assert ("dl3_phase_shift" in names) == (fuse == "0") and ("dl3_shuffle_softmax_xent" in names) == (fuse == "1")
        eng.set_input(x)
        eng.set_targets(y, sw)
        eng.fwd_bwd()
        torch.cuda.synchronize()
        out[fuse] = (eng.grads.cpu().numpy().copy(), float(eng.loss[0].item()), eng.logits().copy())
    assert np.array_equal(out["1"][2], out["0"][2])                     # logits() on demand in the fused engine
    assert abs(out["1"][1] - out["0"][1]) < 1e-6 * abs(out["0"][1])
    assert _l2(out["1"][0], out["0"][0]) < 1e-6


@pytest.mark.parametrize("bn_mode,opt", [("batch", None), ("batch", dict(lr=2e-4, decay=0.5)),
                                         ("frozen", dict(lr=2e-5, decay=0.5, epsilon=1e-7))])
def test_three_train_on_batch_steps_follow_the_adam_oracle(bn_mode, opt):
    """a20, VERDICT r3 #3: the notebook's optimizer — Adam(lr=7e-4, epsilon=1e-8, decay=1e-6), segmentation.ipynb json
    107, Keras 2.2.4 Adam.get_updates with the decay on `iterations` — over THREE Model.train_on_batch calls on three
    different batches (dropout off): every trainable weight, both Adam moments and the BatchNorm moving statistics
    against the float64 oracle's trajectory (oracle/dl3_oracle.py train_steps; an independent torch restatement agrees
    with it to 1e-9 on the CPU, tests/test_oracle.py).  decay = 0.5 makes the schedule itself visible in three steps
    (lr, lr/1.5, lr/2).  Adam turns every gradient into a step of ~lr whatever its size, so elements whose gradient is
    below fp32 rounding noise move by +-lr at random: distances are measured on whole tensors, relative to the update,
    and bounded by twice what the torch restatement's own fp32 run does."""
    from dl3_amd.optimizers import Adam
    from oracle import torch_ref as T
    classes, B, shape = 3, 4, (64, 64, 3)
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    rng = np.random.default_rng(21)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    x0 = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    params = O.calibrate_bn(params, x0, **kw)
    _load(model, params)
    o = dict(O.ADAM_DEFAULTS)
    o.update(opt or {})
    model.compile(optimizer=Adam(**o), loss="sparse_crossentropy_ignoring_last_label", sample_weight_mode="temporal")
    batches = []
    for _ in range(3):
        x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
        y = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
        sw = ((y < classes) * rng.uniform(0.5, 2.0, y.shape)).astype(np.float32)
        batches.append((x, y, sw))
    losses = [model.train_on_batch(x, y[..., None], sw, dropout=False, bn_mode=bn_mode) for x, y, sw in batches]
    eng = model._active
    assert eng.iteration == 3
    eng.sync_all_to_host()
    got = {}
    for l in model.layers:
        got.update(l.weights)
    frozen = bn_mode == "frozen"
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    b64 = [tuple(a.astype(np.float64) for a in b) for b in batches]
    l64, w64, st64 = O.train_steps(p64, b64, opt=o, bn_frozen=frozen, **kw)
    # two fp32 yardsticks: the torch restatement (oneDNN convolutions: close to correctly rounded sums) and the numpy
    # oracle run in float32 (plain fp32 sums).  An activation whose pre-activation lies within rounding noise of 0 or 6
    # flips its ReLU mask, so a gradient's relative error goes with the SQUARE ROOT of the forward error, and Adam then
    # turns a sign into a step of +-lr: the two yardsticks themselves differ by 30x on this trajectory (CPU, round 4:
    # frozen-mode gradients 4.6e-4 / 1.4e-2 median rel-L2, loss of step 2 off by 7e-5 / 2e-3)
    l32, w32 = T.train_steps(params, batches, opt=o, bn_frozen=frozen, dtype=torch.float32, **kw)
    l32n, w32n, _ = O.train_steps(params, batches, opt=o, bn_frozen=frozen, **kw)
    print("losses gpu", losses, "float64", l64, "torch-fp32", l32, "numpy-fp32", l32n)
    for a, b, c, d in zip(losses, l64, l32, l32n):
        assert abs(a - b) <= max(2e-4 * abs(b), 2.0 * max(abs(c - b), abs(d - b))), (losses, l64)
    worst = (0.0, None)
    num = den = n32 = 0.0
    for name, ref in w64.items():
        g = np.asarray(got[name], np.float64).reshape(ref.shape)
        if name.split("/")[-1].startswith("moving"):
            if frozen:
                assert np.array_equal(got[name].reshape(-1), params[name].reshape(-1)), name
            else:
                # (image_pooling_BN sees ONE value per image: the variance of B = 4 nearly equal numbers)
                y32 = max(np.abs(np.asarray(w, np.float64).reshape(ref.shape) - ref).max() for w in (w32[name], w32n[name]))
                assert np.abs(g - ref).max() <= max(2e-4 * max(np.abs(ref).max(), 1e-3), 3.0 * y32), name
            continue
        upd = ref - p64[name]
        if np.abs(upd).max() == 0:
            continue
        d = float(np.sum((g - ref) ** 2))
        d32 = max(float(np.sum((np.asarray(w, np.float64).reshape(ref.shape) - ref) ** 2)) for w in (w32[name], w32n[name]))
        u = float(np.sum(upd ** 2))
        num, den, n32 = num + d, den + u, n32 + d32
        r = np.sqrt(d / u)
        # per tensor: only tensors large enough for a relative L2 to mean something (a handful of flipped signs decide a
        # 32-element beta) and that MOVE in float64 — the beta of a project BatchNorm whose every path leads through a
        # convolution + batch-statistics BatchNorm has an analytically zero gradient (the next BatchNorm removes any
        # per-channel constant): float64 leaves it alone, any fp32 evaluation turns its rounding noise into +-lr steps
        if ref.size >= 1024 and np.sqrt(u / ref.size) >= 0.1 * o["lr"] / (1.0 + 2 * o["decay"]) and r > worst[0]:
            worst = (r, name, np.sqrt(d32 / u))
    tot, tot32 = np.sqrt(num / den), np.sqrt(n32 / den)
    print("update distance to float64, whole model: gpu %.3e, the noisier fp32 yardstick %.3e; worst tensor %s" % (tot, tot32, worst))
    assert tot <= max(2e-2, 2.0 * tot32), (tot, tot32)
    assert worst[1] is not None and worst[0] <= max(0.25, 3.0 * worst[2]), worst
    # the moments themselves, against the same yardstick (they follow the trajectories apart: the weights the second and
    # third gradients are taken at already differ by the sign noise above; the arithmetic of the update itself is pinned
    # to 1e-6 on GIVEN gradients by test_engine_adam_follows_the_oracle_on_given_gradients)
    _, _, st32n = O.train_steps(params, batches, opt=o, bn_frozen=frozen, **kw)
    for name in ("Conv/kernel:0", "expanded_conv_7_depthwise/depthwise_kernel:0", "expanded_conv_16_project_BN/gamma:0",
                 "aspp0/kernel:0", "custom_logits_semantic/bias:0"):
        kind, off, n, shp, _ = eng.slots[name]
        m = eng.adam_m[off:off + n].cpu().numpy().reshape(shp)
        v = eng.adam_v[off:off + n].cpu().numpy().reshape(shp)
        ym, yv = _l2(st32n[name][0], st64[name][0]), _l2(st32n[name][1], st64[name][1])
        em, ev = _l2(m, st64[name][0].reshape(shp)), _l2(v, st64[name][1].reshape(shp))
        print("   moments %-48s m: gpu %.2e numpy-fp32 %.2e   v: gpu %.2e numpy-fp32 %.2e" % (name, em, ym, ev, yv))
        assert em < max(3e-2, 3.0 * ym) and ev < max(3e-2, 3.0 * yv), name


@pytest.mark.parametrize("opt", [None, dict(lr=1e-3, decay=0.5, epsilon=1e-4, beta_1=0.8, beta_2=0.99)])
def test_engine_adam_follows_the_oracle_on_given_gradients(opt):
    """a20: Engine.adam — the host half of Keras 2.2.4 Adam.get_updates (decay on `iterations` BEFORE the increment, t =
    iterations + 1, bias correction folded into lr_t) + dl3_adam_step — over FOUR steps on gradients handed in, against
    oracle/dl3_oracle.adam_update in float64 on the same numbers: weights, m and v to fp32 rounding.  With the
    gradients given there is no trajectory noise: a wrong schedule, epsilon placement or moment update shows at 1e-1,
    the bar is 1e-4 of the update.  (Gradient magnitudes span 1e-9 .. 1e+1: both sides of epsilon.)"""
    model, params = _build("mobilenetv2", (64, 64, 3), 3, "deeplab")
    _load(model, params)
    eng = model._engine(2, True, dropout=False, use_graph=False)
    o = dict(O.ADAM_DEFAULTS)
    o.update(opt or {})
    n = eng.n_param
    rng = np.random.default_rng(31)
    p = eng.params[:n].cpu().numpy().astype(np.float64)
    p0 = p.copy()
    m, v = np.zeros(n), np.zeros(n)
    scale = 10.0 ** rng.uniform(-9, 1, n)
    for it in range(4):
        g = (rng.normal(0, 1, n) * scale).astype(np.float32)
        eng.grads[:n].copy_(torch.from_numpy(g).cuda())
        eng.adam(o, 1.0)
        p, m, v = O.adam_update(p, g.astype(np.float64), m, v, it, **o)
    torch.cuda.synchronize()
    assert eng.iteration == 4
    gp = eng.params[:n].cpu().numpy().astype(np.float64)
    gm, gv = eng.adam_m[:n].cpu().numpy().astype(np.float64), eng.adam_v[:n].cpu().numpy().astype(np.float64)
    # fp32 arithmetic of the update: (1 - beta_2) evaluated in float32 is 0.0010000467 for beta_2 = 0.999f — 4.7e-5 off,
    # in this kernel exactly as in TF's float32 graph; it enters v linearly and the step through 1/sqrt(v)
    assert np.abs(gm - m).max() <= 1e-6 * np.abs(m).max() and _l2(gm, m) < 1e-6
    assert _l2(gv, v) < 1e-4 and np.all(np.abs(gv - v) <= 1e-4 * v + 1e-37)
    upd = p - p0
    # every weight: fp32 representation of the weight itself (4 roundings) + 1e-4 of the four steps it took (each up to
    # ~lr; their signs alternate with the random gradients, so the NET update can be far smaller than the steps)
    assert np.all(np.abs(gp - p) <= 4 * 6e-8 * np.maximum(np.abs(p), np.abs(p0)) + 1e-4 * 4 * o["lr"])
    assert _l2(gp - p0, upd) < 1e-4, _l2(gp - p0, upd)
    # the schedule is visible: the four steps are not four equal steps (decay) and elements below epsilon move less
    big = scale > 1e-2
    small = scale < 1e-8 * max(1.0, o["epsilon"] / 1e-8)
    assert np.abs(upd[big]).mean() > o["lr"] / (1 + 3 * o["decay"]) and np.abs(upd[small]).mean() < 0.5 * np.abs(upd[big]).mean()


def test_notebook_fine_tuning_freezes_the_backbone(monkeypatch):
    """segmentation.ipynb json 147-155 ("fine-tune model (train only last conv layers)"): every layer before
    `concat_projection` gets `trainable = False`.  Keras 2.2.4 semantics: frozen weights never move, a frozen
    BatchNormalization still normalises with batch statistics in the training phase but its moving statistics stay; the
    trainable tail — concat_projection(+BN), the logits convolution — trains as always.  Checked against the oracle's
    train_steps(frozen=...) over two steps, and: the backward pass below the first trainable parameter is not lowered at
    all (round 4), with the same gradients (to fp32 rounding) for what does train."""
    from dl3_amd.optimizers import Adam
    classes, B, shape = 3, 3, (64, 64, 3)
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    rng = np.random.default_rng(41)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    params = O.calibrate_bn(params, rng.integers(0, 256, (B,) + shape).astype(np.float32), **kw)
    _load(model, params)
    flag = 0
    for l in model.layers:              # the notebook's loop, verbatim in effect
        l.trainable = False
        if l.name == "concat_projection":
            flag = 1
        if flag:
            l.trainable = True
    frozen = {n for l in model.layers if not l.trainable for n in l.weights}
    live = {n for l in model.layers if l.trainable for n in l.weights}
    assert "concat_projection/kernel:0" in live and "custom_logits_semantic/bias:0" in live and "aspp0/kernel:0" in frozen
    o = dict(O.ADAM_DEFAULTS, lr=1e-4)
    model.compile(optimizer=Adam(**o))
    batches = []
    for _ in range(2):
        x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
        y = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
        sw = ((y < classes) * rng.uniform(0.5, 2.0, y.shape)).astype(np.float32)
        batches.append((x, y, sw))
    losses = [model.train_on_batch(x, y[..., None], sw, dropout=False) for x, y, sw in batches]
    eng = model._active
    # nothing below concat_projection is differentiated: a handful of launches instead of ~160
    names = [op[0] for op in eng.ops_bwd]
    assert len(names) < 30 and not any(n.startswith("dl3_dwconv3x3") for n in names), names
    assert sum(n.startswith("dl3_pwconv_bwd_data") for n in names) == 1   # logits -> concat_projection's output only
    g_pruned = {n: eng.grad_of(n).copy() for n in live if "/moving_" not in n}
    eng.sync_all_to_host()
    got = {}
    for l in model.layers:
        got.update(l.weights)
    for n in frozen:
        assert np.array_equal(got[n], params[n]), n        # kernels, gamma / beta AND moving statistics of frozen layers
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    b64 = [tuple(a.astype(np.float64) for a in b) for b in batches]
    l64, w64, _ = O.train_steps(p64, b64, opt=o, frozen=frozen, **kw)
    l32, w32, _ = O.train_steps(params, batches, opt=o, frozen=frozen, **kw)
    print("losses gpu", losses, "float64", l64, "numpy-fp32", l32)
    assert abs(losses[0] - l64[0]) < 1e-4 * abs(l64[0])
    assert abs(losses[1] - l64[1]) <= max(2e-4 * abs(l64[1]), 2.0 * abs(l32[1] - l64[1]))
    for n in sorted(live):
        ref = w64[n]
        g = np.asarray(got[n], np.float64).reshape(ref.shape)
        if "/moving_" in n:
            assert np.abs(g - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-3), n
            assert not np.array_equal(got[n], params[n]), n   # a trainable BatchNorm's statistics do move
            continue
        upd = ref - p64[n]
        err, y32 = _l2(g - p64[n], upd), _l2(np.asarray(w32[n], np.float64).reshape(ref.shape) - p64[n], upd)
        print("   %-44s update rel-L2 gpu %.2e numpy-fp32 %.2e" % (n, err, y32))
        assert err <= max(2e-2, 2.0 * y32), (n, err, y32)
    # the same step with the whole backward lowered (DL3_PRUNE_BWD=0): identical gradients for everything that trains
    monkeypatch.setenv("DL3_PRUNE_BWD", "0")
    _load(model, {k: got[k] for k in got})
    full = model._engine(B, True, dropout=False, use_graph=False, seed=77)
    assert len(full.ops_bwd) > 100
    monkeypatch.delenv("DL3_PRUNE_BWD")
    lean = model._engine(B, True, dropout=False, use_graph=False, seed=78)
    x, y, sw = batches[1]
    res = []
    for e in (full, lean):
        e.activate()
        e.sync_all_to_device()
        e.set_input(x)
        e.set_targets(y, sw)
        e.fwd_bwd()
        torch.cuda.synchronize()
        res.append({n: e.grad_of(n).copy() for n in g_pruned})
    # (not bit for bit: with the data gradient lowered the layer's dY is written once by its weight-gradient launch and
    # the per-image column sums are taken of that rounded tensor — one fp32 rounding apart)
    for n in g_pruned:
        assert _l2(res[1][n], res[0][n]) < 1e-5, (n, _l2(res[1][n], res[0][n]))


def test_notebook_order_compile_then_freeze_trains_every_weight():
    """The notebook's LITERAL order (segmentation.ipynb: `model.compile(...)` in cell 2, `l.trainable = False` up to
    concat_projection in cell 5, no recompile): Keras 2.2.4 collected the optimizer's weights at compile(), so every
    weight still trains (it warns about the discrepancy); what the flags do change is the update ops collected when the
    train function is built — the moving statistics of the frozen BatchNormalization layers stay.  And compiling again
    afterwards switches to the pruned fine-tuning plan: the engine is not reused across compile() calls."""
    import warnings
    from dl3_amd.optimizers import Adam
    classes, B, shape = 3, 3, (64, 64, 3)
    model, params = _build("mobilenetv2", shape, classes, "deeplab")
    rng = np.random.default_rng(43)
    kw = dict(backbone="mobilenetv2", input_shape=shape, classes=classes, OS=16, head="deeplab")
    params = O.calibrate_bn(params, rng.integers(0, 256, (B,) + shape).astype(np.float32), **kw)
    _load(model, params)
    o = dict(O.ADAM_DEFAULTS, lr=1e-4)
    model.compile(optimizer=Adam(**o))
    flag = 0
    for l in model.layers:
        l.trainable = False
        if l.name == "concat_projection":
            flag = 1
        if flag:
            l.trainable = True
    frozen_layers = {l.name for l in model.layers if not l.trainable}
    x = rng.integers(0, 256, (B,) + shape).astype(np.float32)
    y = rng.integers(0, classes + 1, (B, shape[0] * shape[1])).astype(np.float32)
    sw = ((y < classes) * rng.uniform(0.5, 2.0, y.shape)).astype(np.float32)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        loss = model.train_on_batch(x, y[..., None], sw, dropout=False)
    assert any("collected trainable weights" in str(w.message) for w in rec)
    eng = model._active
    assert len(eng.ops_bwd) > 100                       # the whole backward pass: nothing is pruned
    eng.sync_all_to_host()
    got = {}
    for l in model.layers:
        got.update(l.weights)
    moved = [n for n in got if "/moving_" not in n and not np.array_equal(got[n], params[n])]
    assert "aspp0/kernel:0" in moved and "Conv/kernel:0" in moved and "concat_projection/kernel:0" in moved
    for l in model.layers:
        for n in l.weights:
            if "/moving_" in n:
                assert np.array_equal(got[n], params[n]) == (l.name in frozen_layers), n
    # one Adam step of the all-trainable oracle on the same batch (the moving statistics aside): the same weights
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    l64, w64, _ = O.train_steps(p64, [(x.astype(np.float64), y.astype(np.float64), sw.astype(np.float64))], opt=o, **kw)
    assert abs(loss - l64[0]) < 1e-4 * abs(l64[0])
    # (Adam's first step is lr * g / (|g| + eps): about lr * sign(g) per element, so an element whose gradient sign fp32
    # cannot resolve costs 2 * lr — the early layers of a batch-statistics step have such elements, the head
    # does not)
    for n, bar in (("aspp0/kernel:0", 0.1), ("concat_projection/kernel:0", 0.1), ("custom_logits_semantic/kernel:0", 0.1),
                   ("expanded_conv_depthwise/depthwise_kernel:0", 0.35)):
        d = _l2(np.asarray(got[n], np.float64).reshape(w64[n].shape) - p64[n], w64[n] - p64[n])
        print("   %-44s update rel-L2 %.3f" % (n, d))
        assert d < bar, (n, d)
    # compile() again with the flags as they are now: the fine-tuning plan, a different engine
    model.compile(optimizer=Adam(**o))
    model.train_on_batch(x, y[..., None], sw, dropout=False)
    lean = model._active
    assert lean is not eng and len(lean.ops_bwd) < 30


def test_batch_feeder_equals_the_host_array_path():
    """feed.BatchFeeder (uint8 images + uint8 label maps from pinned / pageable host memory, H2D on a copy stream, widening
    copy + dl3_prepare_targets on the device, two slots) trains exactly like train_on_batch on the float arrays the
    reference's generator would have built on the host (utils.py:375-402): same losses, same weights, bit for bit —
    through Model.fit_generator(device_feed=True) and through the feeder's own loop."""
    from dl3_amd import utils as U
    classes, B, shape, steps = 5, 2, (64, 64, 3), 5
    rng = np.random.default_rng(51)
    imgs = [rng.integers(0, 256, (B,) + shape, dtype=np.uint8) for _ in range(steps)]
    labs = []
    for _ in range(steps):
        l = rng.integers(0, classes + 1, (B, shape[0], shape[1]), dtype=np.uint8)
        l[l == classes] = 255
        labs.append(l)

    def fresh():
        model, params = _build("mobilenetv2", shape, classes, "deeplab")
        _load(model, params)
        model.compile(optimizer=dict(lr=1e-3))
        return model

    # (a) the host-array path: float32 X, Y / SW as utils.prepare_targets' host twin builds them
    ref = fresh()
    want = []
    for x8, l8 in zip(imgs, labs):
        Y, SW = U.prepare_targets(l8, classes)
        want.append(ref.train_on_batch(x8.astype(np.float32), Y, SW, dropout=False))
    w_ref = ref._active.params.cpu().numpy().copy()

    class Seq:
        def __len__(self):
            return steps

        def __getitem__(self, i):
            return imgs[i], labs[i]

    # (b) the feeder's own loop on an engine with the same settings as (a) (fit_generator's engine has dropout on)
    m2 = fresh()
    from dl3_amd.feed import BatchFeeder
    eng = m2._engine(B, True, dropout=False)
    fd = BatchFeeder(eng, classes, np.uint8)
    losses = []

    def step():
        eng.fwd_bwd()
        eng.adam(dict(lr=1e-3))
        losses.append(eng.loss_handle())

    pinned = [(torch.from_numpy(x).pin_memory(), torch.from_numpy(l).pin_memory()) for x, l in zip(imgs[:3], labs[:3])]
    batches = pinned + list(zip(imgs[3:], labs[3:]))    # pinned tensors (zero copy) and plain numpy arrays (slot buffers)
    assert fd.run(iter(batches), step) == steps
    got = [float(l) for l in losses]
    assert got == want, (got, want)
    assert np.array_equal(eng.params.cpu().numpy(), w_ref)

    # (c) the public entry point runs (dropout on: only finiteness and the step count are checked)
    m3 = fresh()
    hist = m3.fit_generator(Seq(), epochs=2, device_feed=True, n_classes=classes)
    assert len(hist) == 2 * steps and all(np.isfinite(h) for h in hist)
