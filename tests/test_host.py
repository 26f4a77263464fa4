"""CPU tests of the host layer: the Deeplabv3()/SegModel/Subpixel drop-in surface, Keras layer ordering and
naming, weight I/O, and the C-ABI library (loads, exports every symbol of include/dl3.h; no compute calls)."""
import numpy as np
import pytest

import dl3_amd  # noqa: F401
from dl3_amd import capi, graph as G
from dl3_amd.deeplabv3p import Deeplabv3, _make_divisible
from dl3_amd.subpixel import ICNR, Subpixel, icnr_weights
from dl3_amd.utils import Jaccard, SegModel, sparse_accuracy_ignoring_last_label
from oracle import dl3_oracle as O


def test_library_exports_every_header_symbol():
    L = capi.lib()
    protos = capi.protos()
    assert len(protos) >= 30
    for name in protos:
        assert hasattr(L, name), name
    assert L.dl3_version() >= 100
    assert L.dl3_pwconv_partials(65536, 160, 960) > 0
    assert L.dl3_pwconv_bwd_weight_workspace(65536, 160, 960) > 0
    assert L.dl3_dwconv3x3_partials(16, 64, 64, 960, 1, 4, 64, 64, 0) > 0


def test_sizing_queries_of_the_weight_gradient_folds():
    """host-only entry points the engine sizes its fold plan with (no launch): slabs of a 1x1 weight gradient fit its
    workspace; the workgroup count of a fold follows the (rows -> columns per workgroup) rule of dl3_reduce_partials"""
    L = capi.lib()
    for M, K, N in [(8192, 960, 160), (524288, 16, 96), (2, 320, 256), (131072, 256, 21), (65536, 1536, 2048)]:
        for two in (0, 1):
            S = L.dl3_pwconv_bwd_weight_splits(M, K, N, two)
            assert S >= 1 and S * K * N * 4 <= L.dl3_pwconv_bwd_weight_workspace(M, K, N), (M, K, N, two, S)
    assert L.dl3_pwconv_bwd_weight_splits(0, 8, 8, 0) == 0
    # round 4: at 32k-128k rows the 160-wide tiles on a small weight matrix get HALF the M splits (their K x N slabs are a
    # fifth of the launch's bytes); nowhere else, and never more slabs than the workspace query promised
    for K, N in [(160, 960), (960, 160), (576, 160)]:
        few, many = L.dl3_pwconv_bwd_weight_splits(65536, K, N, 1), L.dl3_pwconv_bwd_weight_splits(524288, K, N, 1)
        assert 2 * few <= many + 2 and few >= 8, (K, N, few, many)
        for M in (8192, 16384, 32768, 65536, 131072, 262144, 524288):
            for two in (0, 1):
                S = L.dl3_pwconv_bwd_weight_splits(M, K, N, two)
                assert 1 <= S and S * K * N * 4 <= L.dl3_pwconv_bwd_weight_workspace(M, K, N), (M, K, N, two, S)
    assert L.dl3_pwconv_bwd_weight_splits(65536, 736, 736, 1) == L.dl3_pwconv_bwd_weight_splits(524288, 736, 736, 1)
    # the fused both-gradient kernel: at most 2048 workgroups, at least 512 rows each down to 512 workgroups; slabs fit
    for M, want in [(8388608, 2048), (1048576, 2048), (262144, 512), (131072, 512), (32768, 512), (4096, 128)]:
        S = L.dl3_pwconv_bwd_fused_splits(M, 16, 96)
        assert S == want, (M, S)
        assert S * 16 * 96 * 4 <= L.dl3_pwconv_bwd_fused_workspace(M, 16, 96)
    assert L.dl3_pwconv_bwd_fused_supported(1000, 24, 144) == 2 and L.dl3_pwconv_bwd_fused_supported(1000, 144, 24) == 1
    # round 5: six 32x32 blocks (32 <-> 192); seven or more stay with the two-launch path
    assert L.dl3_pwconv_bwd_fused_supported(1000, 32, 192) == 2 and L.dl3_pwconv_bwd_fused_supported(1000, 192, 32) == 1
    assert L.dl3_pwconv_bwd_fused_supported(1000, 64, 128) == 0 and L.dl3_pwconv_bwd_fused_splits(1000, 64, 128) == 0
    # the weight-stationary forward kernel writes one statistic partial row per workgroup: the partial-row query covers it
    assert L.dl3_pwconv_fwd_impl(2097152, 24, 144) == 1 and L.dl3_pwconv_partials(2097152, 24, 144) >= 768
    assert L.dl3_pwconv_fwd_impl(8192, 24, 144) == 0 and L.dl3_pwconv_fwd_impl(65536, 64, 384) == 0
    # (round 6: from 131 072 rows a reduction of 64 into a 384-wide output takes the MFMA-bound weight-stationary kernel)
    assert L.dl3_pwconv_fwd_impl(2097152, 64, 384) == 2 and L.dl3_pwconv_partials(2097152, 64, 384) >= 85
    for P, n, cols in [(1, 5, 64), (32, 6400, 64), (33, 100, 32), (256, 9 * 960, 32), (257, 9, 8), (1365, 153600, 8)]:
        assert L.dl3_reduce_partials_blocks(P, n) == -(-n // cols)
    assert L.dl3_reduce_partials_blocks(0, 10) == 0 and L.dl3_reduce_partials_blocks(4, 0) == 0


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=2)
    with pytest.raises(capi.DL3Error):
        m.predict(np.zeros((1, 64, 64, 3), np.float32))


def test_constructor_contract():
    with pytest.raises(ValueError):
        Deeplabv3(weights="imagenet")
    with pytest.raises(ValueError):
        Deeplabv3(weights=None, backbone="resnet")
    with pytest.raises(FileNotFoundError):
        Deeplabv3(weights="pascal_voc", input_shape=(64, 64, 3))
    G.clear_session()
    m = Deeplabv3(weights=None, infer=True, input_shape=(64, 64, 3), classes=5)
    assert m.output.shape == (64, 64, 5) and m.name == "deeplabv3p"
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=5)
    assert m.output.shape == (64 * 64, 5)
    # input_tensor path (deeplabv3p.py:260-266, :447-450): the model is built on the caller's Input; like the reference, pooling
    # and resize sizes still come from input_shape (deeplabv3p.py:375,:382), so the two must agree
    G.clear_session()
    inp = G.Input(shape=(96, 64, 3))
    m = Deeplabv3(weights=None, input_tensor=inp, input_shape=(96, 64, 3), classes=4, backbone="xception", OS=8)
    assert m.input is inp and m.output.shape == (96 * 64, 4)


def test_mobilenetv2_structure_goldens():
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(128, 128, 3), classes=2)
    assert m.count_params() == 2141762
    G.clear_session()
    m = Deeplabv3(weights=None, classes=21)
    assert m.count_params() == 2146645 and m.trainable_count() == 2113557
    wl = [l.name for l in m.layers if l.weights]
    assert len(wl) == 109
    assert wl[:4] == ["Conv", "Conv_BN", "expanded_conv_depthwise", "expanded_conv_depthwise_BN"]
    # Keras depth ordering of the ASPP branches
    assert wl[-7:] == ["image_pooling", "image_pooling_BN", "aspp0", "aspp0_BN", "concat_projection",
                       "concat_projection_BN", "logits_semantic"]
    # the linear tail that utils.py:181 cuts at layers[-5]
    assert [l.kind for l in m.layers[-5:]] == ["Dropout", "Conv2D", "ResizeBilinear", "Reshape", "Activation"]
    # OS is silently 8 for mobilenetv2 (deeplabv3p.py:316): 64x64 features at 512x512
    assert m.get_layer("expanded_conv_16_project").output.shape == (64, 64, 320)
    assert m.get_layer("expanded_conv_14_depthwise").cfg["rate"] == 4
    assert m.get_layer("expanded_conv_7_depthwise").cfg["rate"] == 2
    assert m.get_layer("expanded_conv_6_depthwise").cfg["rate"] == 1
    assert m.get_layer("Conv_BN").cfg["momentum"] == 0.999 and m.get_layer("aspp0_BN").cfg["eps"] == 1e-5
    names = set(n for l in m.layers for n in l.weights)
    assert names == set(O.param_shapes("mobilenetv2", 21))
    for l in m.layers:
        for n, w in l.weights.items():
            assert tuple(w.shape) == tuple(O.param_shapes("mobilenetv2", 21)[n]), n


def test_xception_structure_goldens():
    G.clear_session()
    m = Deeplabv3(weights=None, classes=21, backbone="xception", OS=8)
    assert m.count_params() == 41258213 and m.trainable_count() == 41055413
    wl = [l.name for l in m.layers if l.weights]
    assert len(wl) == 293
    i = wl.index("aspp1_depthwise")
    assert wl[i:i + 16] == ["aspp1_depthwise", "aspp2_depthwise", "aspp3_depthwise", "aspp1_depthwise_BN",
                            "aspp2_depthwise_BN", "aspp3_depthwise_BN", "image_pooling", "image_pooling_BN", "aspp0",
                            "aspp1_pointwise", "aspp2_pointwise", "aspp3_pointwise", "aspp0_BN", "aspp1_pointwise_BN",
                            "aspp2_pointwise_BN", "aspp3_pointwise_BN"]
    assert m.get_layer("aspp3_depthwise").cfg["rate"] == 36
    assert m.get_layer("exit_flow_block2_separable_conv3_pointwise").output.shape == (64, 64, 2048)
    assert m.layers[-5].kind == "Activation"  # last decoder ReLU (utils.py:181)
    assert set(n for l in m.layers for n in l.weights) == set(O.param_shapes("xception", 21))
    G.clear_session()
    m16 = Deeplabv3(weights=None, classes=21, backbone="xception", OS=16)
    assert m16.get_layer("aspp3_depthwise").cfg["rate"] == 18
    assert m16.get_layer("exit_flow_block2_separable_conv3_pointwise").output.shape == (32, 32, 2048)


def test_segmodel_heads():
    G.clear_session()
    sm = SegModel(image_size=(512, 512)).create_seg_model("subpixel", n=21)
    assert sm.name == "deeplabv3p_subpixel" and sm.trainable_count() == 2453568
    assert [l.name for l in sm.layers[-3:]] == ["subpixel_1", "reshape_2", "pred_mask"]
    sp = sm.get_layer("subpixel_1")
    k, b = sp.get_weights()
    assert k.shape == (1, 1, 256, 21 * 64) and np.all(b == 0) and sp.output.shape == (512, 512, 21)
    for j in range(21 * 64):  # ICNR mapping survives create_seg_model (utils.py:200-204)
        assert np.array_equal(k[0, 0, :, j], k[0, 0, :, j % 21])
    G.clear_session()
    om = SegModel(image_size=(320, 320)).create_seg_model("original", n=7)
    assert om.name == "deeplabv3p" and om.get_layer("conv_upsample").get_weights()[0].shape == (1, 1, 256, 7)
    assert om.output.shape == (320 * 320, 7)
    G.clear_session()
    xm = SegModel(image_size=(256, 256)).create_seg_model("subpixel", n=21, backbone="xception")
    assert xm.get_layer("subpixel_1").cfg["r"] == 4


def test_icnr_and_make_divisible():
    w = icnr_weights(scale=2, shape=(3, 3, 8, 12))
    assert w.shape == (3, 3, 8, 12)
    X = np.random.default_rng(0).normal(size=(1, 1, 5, 3)).astype(np.float32)
    W = ICNR(lambda shape: X, scale=4)(shape=(1, 1, 5, 48))
    assert np.array_equal(W, O.icnr_from_subkernel(X, 4))
    X3 = np.random.default_rng(1).normal(size=(3, 3, 5, 3)).astype(np.float32)
    assert np.array_equal(ICNR(lambda shape: X3, scale=2)(shape=(3, 3, 5, 12)), O.icnr_from_subkernel(X3, 2))
    for v in (8, 16, 24, 32, 11.2, 33.6, 100):
        assert _make_divisible(v, 8) == O._make_divisible(v, 8)
    assert Subpixel(21, 1, 8).cfg["filters"] == 21 * 64


def test_weight_io_roundtrip(tmp_path):
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=4)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    G.clear_session(seed=5)
    m2 = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=4)
    assert not np.array_equal(m.get_layer("Conv").get_weights()[0], m2.get_layer("Conv").get_weights()[0])
    m2.load_weights(path)  # positional
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    G.clear_session(seed=6)
    m3 = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21)  # logits_semantic != custom_*: skipped by_name
    m3.load_weights(path, by_name=True)
    assert np.array_equal(m.get_layer("aspp0").get_weights()[0], m3.get_layer("aspp0").get_weights()[0])
    with pytest.raises(ValueError):
        m3.get_layer("Conv").set_weights([np.zeros((1, 1, 3, 32), np.float32)])


def test_host_metrics():
    y_true = np.array([[[0], [1], [1], [2]], [[0], [0], [3], [3]]], np.float32)  # 3 = void for C = 3
    probs = np.zeros((2, 4, 3), np.float32)
    for b, row in enumerate([[0, 1, 0, 2], [0, 1, 0, 0]]):
        for i, c in enumerate(row):
            probs[b, i, c] = 1
    assert abs(sparse_accuracy_ignoring_last_label(y_true, probs) - 4 / 6) < 1e-6
    # class 0: img0 1/2, img1 1/4 (predictions on void pixels count in the union) -> 3/8; class 1: 1/2 (img0 only);
    # class 2: 1
    assert abs(Jaccard(y_true, probs) - np.mean([np.mean([1 / 2, 1 / 4]), 1 / 2, 1.0])) < 1e-6
    assert abs(Jaccard(y_true, probs) - O.jaccard(y_true[:, :, 0], probs)) < 1e-12


def _fixture_value(name, shape):
    n = int(np.prod(shape))
    return ((np.arange(n, dtype=np.float32) * 0.25 + len(name)) % 7.0 - 3.0).reshape(shape)


def test_h5lite_reads_a_real_h5py_file():
    """tests/golden/keras_style_h5py.h5 was written by h5py/libhdf5 (make_h5_fixture.py) in the Keras 2.2.x layout"""
    import os
    from dl3_amd import h5lite
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "keras_style_h5py.h5")
    names, per = h5lite.read_keras_weights(path)
    assert len(names) == 46 and names[:4] == ["input_1", "lambda_1", "Conv", "Conv_BN"]
    assert per["input_1"] == [] and [w for w, _ in per["Conv_BN"]] == [
        "Conv_BN/gamma:0", "Conv_BN/beta:0", "Conv_BN/moving_mean:0", "Conv_BN/moving_variance:0"]
    n = 0
    for l in names:
        for w, a in per[l]:
            assert a.dtype == np.float32 and np.array_equal(a, _fixture_value(w, a.shape)), w
            n += 1
    assert n == 48
    r = h5lite.Reader(path)
    a = r.attrs(r.root["ohdr"])
    assert a["backend"] == b"tensorflow" and a["keras_version"] == b"2.2.4"


def test_h5_weight_roundtrip_through_h5lite(tmp_path, monkeypatch):
    """model.save_weights('x.h5') / load_weights('x.h5', by_name) with the package's own HDF5 writer/reader"""
    from dl3_amd import h5io, h5lite
    monkeypatch.setattr(h5io, "_h5py", lambda: None)  # force the self-contained path even if h5py exists
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21)
    path = str(tmp_path / "deeplabv3_mobilenetv2_tf_dim_ordering_tf_kernels.h5")
    m.save_weights(path)
    names, per = h5lite.read_keras_weights(path)
    assert names == [l.name for l in m.layers]
    G.clear_session(seed=3)
    m2 = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21)
    m2.load_weights(path, by_name=True)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    # weights='pascal_voc' resolves the bonlime file name locally (deeplabv3p.py:456-465 without the download)
    monkeypatch.setenv("DL3_WEIGHTS_DIR", str(tmp_path))
    G.clear_session(seed=4)
    m3 = Deeplabv3(weights="pascal_voc", input_shape=(64, 64, 3), classes=21)
    assert np.array_equal(m3.get_layer("aspp0").get_weights()[0], m.get_layer("aspp0").get_weights()[0])
    with pytest.raises(h5lite.H5Error):
        bad = tmp_path / "bad.h5"
        bad.write_bytes(b"not hdf5 at all")
        h5lite.read_keras_weights(str(bad))


def test_crf_hook_parameters():
    """the Dense-CRF post-process is a host hook; its parameters are the reference's (utils.py:78-86)"""
    from dl3_amd import utils as U
    assert U.CRF_PARAMS == dict(gt_prob=0.7, gaussian_sxy=(3, 3), gaussian_compat=3, bilateral_sxy=80, bilateral_srgb=13,
                                bilateral_compat=10, iterations=5)
    try:
        import pydensecrf  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="pydensecrf"):
            U.do_crf(np.zeros((8, 8, 3), np.uint8), np.zeros((8, 8), np.int32))


def test_crf_label_restore_reproduces_the_reference_loop():
    """utils.py:86-89 maps MAP indices back with an in-place np.putmask loop over ascending indices; a value written for
    an earlier index is matched again by a later one.  SURVEY G5: reproduce, do not fix."""
    from dl3_amd import utils as U
    colors = np.array([0, 2, 15])
    MAP = np.array([[0, 1, 2], [1, 1, 0], [2, 0, 1]])
    got = U.restore_crf_labels(MAP.copy(), colors)
    # index 1 -> 2, then every 2 (old index 2 AND the freshly written ones) -> 15: class 2 vanishes
    assert np.array_equal(got, np.array([[0, 15, 15], [15, 15, 0], [15, 0, 15]]))
    # contiguous label values 0..n-1 come back unchanged, and so do values that never collide with a later index
    assert np.array_equal(U.restore_crf_labels(MAP.copy(), np.array([0, 1, 2])), MAP)
    assert np.array_equal(U.restore_crf_labels(MAP.copy(), np.array([7, 9, 30])), np.array([7, 9, 30])[MAP])
    # an index absent from MAP is never touched: colors {0, 2, 15} with no pixel of index 1 keep class 2's pixels ... gone too
    assert np.array_equal(U.restore_crf_labels(np.array([0, 2, 2]), colors), np.array([0, 15, 15]))


def test_crf_hook_runs_against_a_recording_pydensecrf(monkeypatch):
    """N4 executed: do_crf driven end to end against a stand-in `pydensecrf` that records every call (the real package
    is not installable offline).  Checks the call sequence, argument values and shapes the reference issues
    (utils.py:74-91: DenseCRF2D(width, height, n), unary_from_labels(labels, n, gt_prob=.7, zero_unsure), Gaussian sxy
    (3,3) compat 3, bilateral sxy 80 srgb 13 compat 10 on the uint8 image, 5 iterations) and the label restore."""
    import sys
    import types
    from dl3_amd import utils as U
    calls = []

    class DenseCRF2D:
        def __init__(self, w, h, n):
            calls.append(("init", w, h, n))
            self.n, self.hw = n, h * w

        def setUnaryEnergy(self, u):
            calls.append(("unary", u.shape, u.dtype))
            self.u = u

        def addPairwiseGaussian(self, sxy, compat):
            calls.append(("gaussian", sxy, compat))

        def addPairwiseBilateral(self, sxy, srgb, rgbim, compat):
            calls.append(("bilateral", sxy, srgb, rgbim.dtype, rgbim.shape, rgbim.flags["C_CONTIGUOUS"], compat))

        def inference(self, it):
            calls.append(("inference", it))
            return np.exp(-self.u)  # the unary alone: MAP = the input labels

    def unary_from_labels(labels, n, gt_prob, zero_unsure):
        calls.append(("unary_from_labels", labels.shape, n, gt_prob, zero_unsure))
        u = np.full((n, labels.size), -np.log((1.0 - gt_prob) / (n - 1)), np.float32)
        u[labels.reshape(-1), np.arange(labels.size)] = -np.log(gt_prob)
        return u

    pkg, dm, um = types.ModuleType("pydensecrf"), types.ModuleType("pydensecrf.densecrf"), types.ModuleType("pydensecrf.utils")
    dm.DenseCRF2D, um.unary_from_labels = DenseCRF2D, unary_from_labels
    pkg.densecrf, pkg.utils = dm, um
    for k, v in (("pydensecrf", pkg), ("pydensecrf.densecrf", dm), ("pydensecrf.utils", um)):
        monkeypatch.setitem(sys.modules, k, v)
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (6, 9, 3)).astype(np.float32)
    mask = rng.integers(0, 3, (6, 9)).astype(np.int32)
    out = U.do_crf(im, mask, zero_unsure=False)
    assert calls[0] == ("init", 9, 6, 3)
    assert calls[1] == ("unary_from_labels", (54,), 3, 0.7, False)
    assert calls[2][0] == "unary" and calls[2][1] == (3, 54)
    assert calls[3] == ("gaussian", (3, 3), 3)
    assert calls[4] == ("bilateral", 80, 13, np.dtype("uint8"), (6, 9, 3), True, 10)
    assert calls[5] == ("inference", 5)
    assert out.shape == (6, 9) and np.array_equal(out, mask)
    # non-contiguous original values go through the reference's in-place restore (class 2 -> 15 together with index 2)
    out2 = U.do_crf(im, np.array([0, 2, 15])[mask], zero_unsure=True)
    assert np.array_equal(out2, np.array([0, 15, 15])[mask])


def test_stored_channel_plan():
    """engine.plan_channels: Xception's 728-channel tensors are stored 736 wide (whole 128-byte lines per pixel row),
    nothing else changes; device weight shapes follow; the slices of a Concatenate keep their logical widths"""
    from dl3_amd import engine as E
    assert [E.stored_channels(c) for c in (16, 24, 144, 304, 728, 960, 1024, 2048)] == [16, 24, 144, 304, 736, 960, 1024, 2048]
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21, backbone="mobilenetv2", OS=16)
    phys = E.plan_channels(m)
    assert all(phys[id(l.output)] == l.output.shape[-1] for l in m.layers)
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=21, backbone="xception", OS=8)
    phys = E.plan_channels(m)
    L = {l.name: l for l in m.layers}
    wide = [l.name for l in m.layers if phys[id(l.output)] != l.output.shape[-1]]
    assert wide and all(L[n].output.shape[-1] == 728 and phys[id(L[n].output)] == 736 for n in wide)
    for n in ("entry_flow_block3_separable_conv1_pointwise", "middle_flow_unit_7_separable_conv2_depthwise_BN",
              "exit_flow_block1_separable_conv1_depthwise", "entry_flow_block3_shortcut", "add_3"):
        if n in L:
            assert n in wide, n
    l = L["middle_flow_unit_1_separable_conv1_pointwise"]
    assert E.device_shape(phys, l, l.name + "/kernel:0", (1, 1, 728, 728)) == (1, 1, 736, 736)
    l = L["middle_flow_unit_1_separable_conv1_depthwise"]
    assert E.device_shape(phys, l, l.name + "/depthwise_kernel:0", (3, 3, 728, 1)) == (3, 3, 736, 1)
    l = L["exit_flow_block1_separable_conv2_pointwise"]  # 728 -> 1024: only the input side is widened
    assert E.device_shape(phys, l, l.name + "/kernel:0", (1, 1, 728, 1024)) == (1, 1, 736, 1024)
    l = L["middle_flow_unit_1_separable_conv1_pointwise_BN"]
    assert E.device_shape(phys, l, l.name + "/gamma:0", (728,)) == (736,)
    for l in m.layers:
        if l.kind == "Concatenate":
            assert phys[id(l.output)] == l.output.shape[-1]
            assert all(phys[id(t)] == t.shape[-1] for t in l.inbound)


def test_adam_shim_and_compile_surface():
    """segmentation.ipynb cell 2 compiles with keras.optimizers.Adam(lr=7e-4, epsilon=1e-8, decay=1e-6): compile takes
    the object, a dict, 'adam' or None — and rejects everything else AT compile time"""
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    from dl3_amd.optimizers import Adam, as_adam_dict
    a = Adam(lr=7e-4, epsilon=1e-8, decay=1e-6)
    assert a.get_config() == dict(lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6, amsgrad=False)
    assert Adam().get_config()["epsilon"] == 1e-7 and Adam().lr == 0.001      # Keras 2.2.4 defaults
    assert Adam.from_config(a.get_config()).get_config() == a.get_config()
    assert as_adam_dict(None) == {} and as_adam_dict("adam")["lr"] == 0.001
    assert as_adam_dict(dict(lr=1e-3)) == {"lr": 1e-3}
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=3, backbone="mobilenetv2")
    m.compile(optimizer=a, sample_weight_mode="temporal", loss="whatever", metrics=[])
    assert m._compiled["optimizer"] == dict(lr=7e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=1e-6)
    assert m._compiled["optimizer_object"] is a
    for bad in (object(), 3.0, "sgd", dict(learning_rate=1.0)):
        with pytest.raises((TypeError, ValueError)):
            m.compile(optimizer=bad)
    with pytest.raises(ValueError):
        Adam(amsgrad=True)

    class SGD:  # a Keras-style optimizer that is not Adam
        def get_config(self):
            return dict(lr=0.1)
    with pytest.raises(ValueError):
        m.compile(optimizer=SGD())


def test_subpixel_constructor_contract():
    """subpixel.py:42-58: any square kernel_size builds a [k,k,Cin,r*r*filters] kernel (the engine lowers it through the
    k x k tap gather, tests/test_gpu_model.py::test_subpixel_with_kernel_size_above_one); what is not on the path raises"""
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.subpixel import Subpixel
    G.clear_session()
    t = G.Input(shape=(8, 8, 5))
    l = Subpixel(3, 3, 2, padding="same", name="sp3")
    out = l(t)
    assert out.shape == (16, 16, 3) and l.weights["sp3/kernel:0"].shape == (3, 3, 5, 12)
    out = Subpixel(3, (2, 2), 2, padding="valid", name="sp2")(G.Input(shape=(8, 8, 5)))
    assert out.shape == (14, 14, 3)
    for kw in (dict(strides=(2, 2)), dict(activation="relu"), dict(padding="causal")):
        with pytest.raises(ValueError):
            Subpixel(3, 3, 2, **kw)
    with pytest.raises(ValueError):
        Subpixel(3, (3, 1), 2)


def test_trainable_flags_follow_keras_224_compile_semantics():
    """Keras 2.2.4: the optimizer's weight list is collected at compile(), the layers' update ops when the train function
    is first built; `layer.trainable` flipped after compile() (segmentation.ipynb: compile in cell 2, the fine-tuning loop
    in cell 5) only stops the moving statistics of the frozen BatchNormalization layers — with a warning —, and a plan
    lowered for one set of flags is never reused for another.  Host logic only: no engine is built."""
    import warnings
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=3, backbone="mobilenetv2")

    def freeze():
        flag = 0
        for l in m.layers:
            l.trainable = False
            if l.name == "concat_projection":
                flag = 1
            if flag:
                l.trainable = True

    # never compiled: the live flags, and the key changes with them
    opt0, upd0, key0 = m._training_flags()
    assert all(opt0.values()) and opt0 == upd0 and key0[0] == "live"
    freeze()
    opt1, upd1, key1 = m._training_flags()
    assert not opt1["aspp0"] and opt1["concat_projection"] and key1 != key0
    for l in m.layers:
        l.trainable = True
    # compile, THEN freeze (the notebook's order): every weight still trains, frozen layers' updates stop, Keras' warning
    m.compile(optimizer=None)
    freeze()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        opt2, upd2, key2 = m._training_flags()
    assert any("collected trainable weights" in str(w.message) for w in rec)
    assert all(opt2.values()) and not upd2["aspp0_BN"] and upd2["concat_projection_BN"] and key2 == ("compiled", 1)
    for l in m.layers:   # flipping flags again without compile() changes nothing: the train function exists
        l.trainable = True
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert m._training_flags() == (opt2, upd2, key2) and not rec
    # freeze, then compile: the tail only; a new key, so the old plan is not reused
    freeze()
    m.compile(optimizer=None)
    opt3, upd3, key3 = m._training_flags()
    assert not opt3["aspp0"] and opt3["concat_projection"] and opt3 == upd3 and key3 == ("compiled", 2)


def test_keras_split_attributes_are_read_and_written(tmp_path):
    """Keras 2.2.4 stores a `layer_names` / `weight_names` attribute larger than 64 512 bytes in pieces `name0`, `name1`, ...
    (engine/saving.py save_attributes_to_hdf5_group): both readers reassemble them, the writer cuts
    them the way Keras does.  Forced here with a small limit on a real model file."""
    import dl3_amd  # noqa: F401
    from dl3_amd import graph as G, h5io, h5lite
    from dl3_amd.deeplabv3p import Deeplabv3
    # the helpers on their own: one attribute below the limit, the smallest number of equal pieces above it
    names = ["layer_%03d" % i for i in range(100)]
    assert h5io.attr_pieces("layer_names", names) == [("layer_names", [n.encode() for n in names])]
    pieces = h5io.attr_pieces("layer_names", names, limit=300)
    assert [a for a, _ in pieces] == ["layer_names%d" % i for i in range(len(pieces))] and len(pieces) == 4
    assert all(len(p) * 9 <= 300 for _, p in pieces) and [x for _, p in pieces for x in p] == [n.encode() for n in names]
    as_attrs = {a: np.array(p, dtype="S") for a, p in pieces}
    assert h5io.attr_list(as_attrs, "layer_names") == names                   # what the h5py reader does with .attrs
    assert h5io.attr_list({"layer_names": np.array(names, "S")}, "layer_names") == names
    assert h5io.attr_list({}, "layer_names") == []
    # a real model through the package's own HDF5 writer / reader with every string attribute in pieces
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=3, backbone="mobilenetv2")
    layers = [(l.name, list(zip(l.weights.keys(), l.get_weights()))) for l in m.layers]
    path = str(tmp_path / "split.h5")
    h5lite.write_keras_weights(path, layers, attr_limit=400)
    r = h5lite.Reader(path)
    root_attrs = r.attrs(r.root["ohdr"])
    assert "layer_names" not in root_attrs and "layer_names0" in root_attrs and "layer_names3" in root_attrs
    got_names, per = h5lite.read_keras_weights(path)
    assert got_names == [n for n, _ in layers]
    for n, ws in layers:
        assert [k for k, _ in per[n]] == [k for k, _ in ws]
        for (_, a), (_, b) in zip(per[n], ws):
            assert np.array_equal(a, b)
    # and the model loads it by name
    G.clear_session()
    m2 = Deeplabv3(weights=None, input_shape=(64, 64, 3), classes=3, backbone="mobilenetv2")
    m2.load_weights(path, by_name=True)
    for la, lb in zip(m.layers, m2.layers):
        for a, b in zip(la.get_weights(), lb.get_weights()):
            assert np.array_equal(a, b)


def _dry_engine(monkeypatch, backbone, B, size=512, OS=16, **kw):
    """lower a plan WITHOUT a GPU: Engine only records launches and asks libdl3.so's host-side sizing queries while it
    lowers; with torch.cuda.is_available patched and the buffers on the CPU (untouched virtual memory) the whole plan —
    ops, arguments, workspace sizes, consistency checks — can be inspected here"""
    import collections
    import torch
    import dl3_amd  # noqa: F401
    from dl3_amd import capi, graph as G
    from dl3_amd.deeplabv3p import Deeplabv3
    from dl3_amd.engine import Engine
    capi.lib()
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    G.clear_session()
    m = Deeplabv3(weights=None, input_shape=(size, size, 3), classes=21, backbone=backbone, OS=OS)
    e = Engine(m, batch=B, training=True, device="cpu", **kw)
    return e, collections.Counter(op[0] for op in e.ops_fwd + e.ops_bwd)


def test_knob_table_is_current():
    """KNOBS.md (tools/knobs.py) lists exactly the DL3_* environment variables the sources read — 23 since round 6 removed
    the settled tuning aids — each with default, kind and meaning"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "knobs.py"), "--check"], capture_output=True, text=True)
    assert res.returncode == 0, "KNOBS.md is stale or a knob lacks a description: run python tools/knobs.py\n" + res.stdout
    text = open(os.path.join(root, "KNOBS.md")).read()
    assert text.count("| `DL3_") == 25


def test_default_kernel_routes_by_name(monkeypatch):
    """the routes of the benchmarked plans BY NAME (dl3_pwconv_route): what bench.py times is what this asserts.
    cfg2 at B=128: the HBM-bound early layers on the weight-stationary streaming kernel, the 64 x 64 blocks' expand
    convolutions (reduction 64 / 96 / 160) forward and their project convolutions' bwd-data on the MFMA-bound weight-stationary
    kernel, the expand convolutions' weight gradients on the one-tile-row kernel, everything else tiled; B=16: no
    weight-stationary MFMA kernel (below 131 072 rows); B=2: the K-split kernel on the 960 <-> 160 / 576 <-> 96 layers."""
    from dl3_amd import capi
    L = capi.lib()
    R = dict(tiled=0, ws_hbm=1, ws_mfma=2, ksplit=3, wgrad_row=4, narrow=5)
    M = 128 * 64 * 64
    # the logits layer (deeplabv3p.py:438): forward from B=2, bwd-data from B=4, weight gradient from B=32
    assert [L.dl3_pwconv_route(d, M, 256, 21) for d in (0, 3, 4)] == [R["narrow"]] * 3
    assert [L.dl3_pwconv_route(d, 16 * 64 * 64, 256, 21) for d in (0, 3, 4)] == [R["narrow"], R["narrow"], R["tiled"]]
    assert [L.dl3_pwconv_route(d, 2 * 64 * 64, 256, 21) for d in (0, 3, 4)] == [R["narrow"], R["tiled"], R["tiled"]]
    for K, N in ((160, 960), (96, 576), (64, 384)):
        assert L.dl3_pwconv_route(0, M, K, N) == R["ws_mfma"] and L.dl3_pwconv_fwd_impl(M, K, N) == 2
        assert L.dl3_pwconv_route(1, M, N, K) == R["ws_mfma"]         # bwd-data of the project convolution N -> K
        assert L.dl3_pwconv_route(0, 16 * 64 * 64, K, N) == R["tiled"]  # B=16: 65 536 rows — forward tiled, ...
        assert L.dl3_pwconv_route(1, 16 * 64 * 64, N, K) == (R["ws_mfma"] if K != 64 else R["tiled"])   # ... bwd-data from 65 536
        assert L.dl3_pwconv_route(0, 24 * 64 * 64, K, N) == (R["ws_mfma"] if K != 64 else R["tiled"])   # forward from 98 304
        assert L.dl3_pwconv_route(1, 8 * 64 * 64, N, K) == R["tiled"]
    for K, N in ((960, 160), (576, 96), (384, 64), (960, 320), (320, 256), (256, 256)):
        assert L.dl3_pwconv_route(0, M, K, N) == R["tiled"] and L.dl3_pwconv_route(1, M, N, K) == R["tiled"]
    assert L.dl3_pwconv_route(0, 128 * 256 * 256, 16, 96) == R["ws_hbm"] and L.dl3_pwconv_route(0, 128 * 128 * 128, 24, 144) == R["ws_hbm"]
    assert L.dl3_pwconv_route(2, M, 160, 960) == R["wgrad_row"] and L.dl3_pwconv_route(2, M, 96, 576) == R["wgrad_row"]
    assert L.dl3_pwconv_route(2, M, 960, 160) == R["tiled"] and L.dl3_pwconv_route(2, M, 64, 384) == R["tiled"]
    assert L.dl3_pwconv_route(0, 2 * 64 * 64, 960, 160) == R["ksplit"] and L.dl3_pwconv_route(0, 2 * 64 * 64, 576, 96) == R["ksplit"]
    assert L.dl3_pwconv_route(0, 2 * 64 * 64, 160, 960) == R["tiled"]
    # ... and through a lowered plan: the launches of the B=128 engine that take each route
    e, c = _dry_engine(monkeypatch, "mobilenetv2", 128)
    fwd = [L.dl3_pwconv_route(0, op[2][9], op[2][10], op[2][11]) for op in e.ops_fwd if op[0] == "dl3_pwconv_fwd"]
    assert fwd.count(R["ws_mfma"]) == 10 and fwd.count(R["ws_hbm"]) == 12 and fwd.count(R["narrow"]) == 1
    wg = [L.dl3_pwconv_route(2, op[2][14], op[2][15], op[2][16]) for op in e.ops_bwd if op[0] == "dl3_pwconv_bwd_weight_dy"]
    assert wg.count(R["wgrad_row"]) == 6
    # the statistic partial rows the engine sized cover the weight-stationary kernel's one row per row group
    assert L.dl3_pwconv_partials(M, 160, 960) >= 42 and L.dl3_pwconv_partials(M, 960, 160) >= 42


def test_benchmarked_plan_lowers_consistently_without_a_gpu(monkeypatch):
    """the plan bench.py times (cfg2, B=128) and its small-batch / frozen / data-parallel / Xception relatives, lowered on
    the CPU: kernel routes of round 5 (12 both-gradient launches incl. the six-block 32 <-> 192 layers, 12 forward
    launches on the weight-stationary kernel), ONE dY buffer with writer and reader adjacent (Engine._check_dy_adjacency),
    centred frozen BatchNorm never read raw (Engine._check_centred_consumers), and the data-parallel engine's count / loss
    sum living in the arena tail"""
    from dl3_amd import capi
    L = capi.lib()
    e, c = _dry_engine(monkeypatch, "mobilenetv2", 128)
    assert c["dl3_pwconv_bwd_fused"] == 12 and c["dl3_pwconv_bwd_weight_dy"] == 20
    ws = [op for op in e.ops_fwd if op[0] == "dl3_pwconv_fwd" and L.dl3_pwconv_fwd_impl(op[2][9], op[2][10], op[2][11]) == 1]
    assert len(ws) == 12 and {(op[2][10], op[2][11]) for op in ws} == {(32, 16), (16, 96), (96, 24), (24, 144), (144, 24),
                                                                      (144, 32), (32, 192), (192, 32)}
    for op in ws:   # the statistic partial buffer the engine sized covers the kernel's one row per workgroup
        assert L.dl3_pwconv_partials(op[2][9], op[2][10], op[2][11]) >= 768
    assert e.dy_buf is not None and c["dl3_count_nonzero"] == 1
    e2, c2 = _dry_engine(monkeypatch, "mobilenetv2", 2)
    assert c2["dl3_pwconv_bwd_fused"] == 6            # only the 256x256 / 128x128 layers have >= 32768 rows at B=2
    # frozen BatchNorm: statistics launches gone, every BatchNorm'ed 1x1 output centred and never read raw
    e3, c3 = _dry_engine(monkeypatch, "mobilenetv2", 2, bn_mode="frozen")
    assert c3["dl3_bn_finalize"] == 0 and c3["dl3_bn_frozen_centered"] == 0   # (prep ops are not in fwd / bwd)
    assert sum(1 for op in e3.ops_prep if op[0] == "dl3_bn_frozen_centered") >= 35
    assert any(b.centred for b in e3.bufs)
    # data parallel: the shard's count(w != 0) and loss sum are written INTO the gradient arena's tail
    e4, c4 = _dry_engine(monkeypatch, "mobilenetv2", 2, external_nnz=True)
    cnt = [op for op in e4.ops_fwd if op[0] == "dl3_count_nonzero"][0]
    assert cnt[2][2] == e4.grads.data_ptr() + 4 * e4.tail and e4.grads.numel() == e4.tail + 4
    red = [op for op in e4.ops_fwd if op[0] == "dl3_reduce_partials"][-1]
    assert red[2][3] == e4.grads.data_ptr() + 4 * (e4.tail + 1) == e4.loss.data_ptr()
    assert e4.nnz_host == 512 * 512 and float(e4.nnz[0]) == 512 * 512
    # Xception OS=8: no fused launches (no layer small enough), dY materialised for its 1x1 convolutions
    e5, c5 = _dry_engine(monkeypatch, "xception", 1, size=256, OS=8)
    assert c5["dl3_pwconv_bwd_fused"] == 0 and c5["dl3_pwconv_bwd_weight_dy"] > 50
