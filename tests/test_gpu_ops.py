"""-m gpu parity tests: every libdl3.so operator, called through the C ABI, against the numpy oracle
(oracle/dl3_oracle.py) on the same seeded inputs.  fp32 tolerance: max|a-b| <= TOL * max|b|."""
import numpy as np
import pytest
import torch

from oracle import dl3_oracle as O
from tests.gpu_util import (call, dev, dropout_keep_mask, empty, fold_partials, host, np_act, np_mask, ptr, relerr,
                            release, stream)

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def L():
    import dl3_amd  # noqa: F401
    from dl3_amd import capi
    return capi.lib()


def _xform(rng, C, act):
    if act is None:
        return None, None, 0
    s = rng.uniform(0.5, 1.5, C).astype(np.float32)
    t = rng.normal(0, 0.5, C).astype(np.float32)
    return s, t, act


DW_CASES = [
    # N, H, W, C, stride, rate, (pad_t, pad_l) or None=SAME, impl, act
    (2, 16, 16, 32, 1, 1, None, 0, None),
    (2, 16, 16, 96, 1, 2, None, 0, 2),
    (1, 32, 32, 64, 1, 4, None, 0, 2),
    (2, 17, 19, 40, 1, 3, None, 0, 1),
    (2, 16, 16, 32, 1, 2, None, 1, 2),   # gather impl on a march-able shape
    (2, 16, 16, 32, 2, 1, None, 0, 2),   # SAME stride 2, even size: pad (0,0)
    (2, 15, 15, 24, 2, 1, None, 0, 2),   # SAME stride 2, odd size: pad (1,1)
    (2, 16, 16, 32, 2, 1, (1, 1), 0, 1),  # explicit ZeroPadding2D + VALID (Xception)
    (1, 8, 8, 16, 1, 12, None, 0, 1),    # rate > size
    (3, 64, 64, 144, 1, 1, None, 0, 2),  # C not a multiple of 32, several row chunks
    (2, 8, 8, 64, 1, 2, None, 0, 2),     # map narrower than the 32 pixel lanes (idle lanes must stay in bounds)
    (1, 4, 4, 32, 1, 1, None, 0, 1),
    (16, 36, 36, 512, 1, 18, None, 0, 2),  # large rate: several row phases per workgroup (ASPP rates 12/24/36)
    (2, 24, 64, 96, 1, 4, None, 0, 2),     # forward with two pixels per lane: one 64-pixel segment, rate 4
    (1, 20, 96, 40, 1, 2, None, 0, 1),     # ... one and a half segments, C not a multiple of 32
    (2, 9, 33, 32, 1, 1, None, 0, 2),      # ... odd width
    (1, 40, 80, 64, 1, 8, None, 0, None),  # ... rate 8
    # Xception OS=8 ASPP (deeplabv3p.py:273-277,:390-399): the 64x64x2048 map of cfg4 at rates 12 / 24 / 36 — side taps
    # live (unlike the 8x8 map of the small model test), several row phases per workgroup
    (1, 64, 64, 2048, 1, 12, None, 0, 1),
    (1, 64, 64, 2048, 1, 24, None, 0, 1),
    (1, 64, 64, 2048, 1, 36, None, 0, 1),
    (1, 64, 64, 728, 1, 2, None, 0, 1),    # middle flow at OS=8: 728 channels (22.75 slabs of 32), rate 2
    (1, 33, 33, 1536, 1, 4, None, 0, 1),   # exit flow rate 4, odd map
    # the two-pixel FORWARD plan has more partial rows than the backward plan (96 / 48 against 72 / 36): the backward
    # launch must zero the rows of the NaN-filled partial buffers it does not own
    (3, 33, 65, 960, 1, 16, None, 0, 2),
    (3, 33, 65, 960, 1, 2, None, 0, 2),
]


def _dw_geom(H, W, stride, rate, pads):
    if pads is None:
        Ho, pt, _ = O.same_pads(H, 3, stride, rate)
        Wo, pl, _ = O.same_pads(W, 3, stride, rate)
    else:
        pt, pl = pads
        Ho = (H + 2 * pt - ((3 - 1) * rate + 1)) // stride + 1
        Wo = (W + 2 * pl - ((3 - 1) * rate + 1)) // stride + 1
    return Ho, Wo, pt, pl


@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_fwd(L, case):
    N, H, W, C, stride, rate, pads, impl, act = case
    rng = np.random.default_rng(0)
    Ho, Wo, pt, pl = _dw_geom(H, W, stride, rate, pads)
    x = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, C)).astype(np.float32)
    s, t, a = _xform(rng, C, act)
    xin = x if s is None else np_act(s * x + t, a)
    ref = O.depthwise3x3(xin.astype(np.float64), w.astype(np.float64), stride, rate, pt, pl, Ho, Wo)
    P = L.dl3_dwconv3x3_partials(N, H, W, C, stride, rate, Ho, Wo, impl)
    y, part = empty(N, Ho, Wo, C), empty(P, C, 2)
    call("dl3_dwconv3x3_fwd", ptr(dev(x)), ptr(dev(s)) if s is not None else None,
         ptr(dev(t)) if t is not None else None, a, ptr(dev(w)), ptr(y), N, H, W, C, stride, rate, pt, pl, Ho, Wo,
         ptr(part), impl)
    assert relerr(host(y), ref) < TOL
    s1, s2 = fold_partials(part, P, C)
    assert relerr(s1, ref.sum((0, 1, 2))) < 1e-3
    assert relerr(s2, (ref ** 2).sum((0, 1, 2))) < 1e-3


@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_bwd(L, case):
    N, H, W, C, stride, rate, pads, impl, act = case
    rng = np.random.default_rng(1)
    Ho, Wo, pt, pl = _dw_geom(H, W, stride, rate, pads)
    x = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, C)).astype(np.float32)
    g = rng.normal(0, 1, (N, Ho, Wo, C)).astype(np.float32)
    yraw = rng.normal(0, 1, (N, Ho, Wo, C)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, C).astype(np.float32) for _ in range(3)]
    add = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    mean = rng.normal(0, 1, C).astype(np.float32)
    invstd = rng.uniform(0.5, 2, C).astype(np.float32)
    s, t, a = _xform(rng, C, act)
    z = x.astype(np.float64) if s is None else (s * x.astype(np.float64) + t)
    xin = np_act(z, a)
    dY = cA * g.astype(np.float64) + cB * yraw + cC
    tape = O.Tape()
    wv = w.astype(np.float64)
    yref = O.depthwise3x3(xin, wv, stride, rate, pt, pl, Ho, Wo, tape=tape)
    grads = tape.backward(yref, dY)
    dx_ref = grads[id(xin)] * np_mask(z, a) + add
    dw_ref = grads[id(wv)]
    P = L.dl3_dwconv3x3_partials(N, H, W, C, stride, rate, Ho, Wo, impl)
    dx, dpart, wpart = empty(N, H, W, C), empty(P, C, 2), empty(P, 9, C)
    call("dl3_dwconv3x3_bwd", ptr(dev(g)), ptr(dev(yraw)), ptr(dev(cA)), ptr(dev(cB)), ptr(dev(cC)), ptr(dev(x)),
         ptr(dev(s)) if s is not None else None, ptr(dev(t)) if t is not None else None, a, ptr(dev(w)), ptr(dx),
         ptr(dev(add)), ptr(dev(mean)), ptr(dev(invstd)), ptr(dpart), ptr(wpart), N, H, W, C, stride, rate, pt, pl,
         Ho, Wo, impl)
    assert relerr(host(dx), dx_ref) < TOL
    dwg = host(wpart).reshape(P, 3, 3, C).astype(np.float64).sum(0)
    assert relerr(dwg, dw_ref) < 1e-3
    s1, s2 = fold_partials(dpart, P, C)
    assert relerr(s1, dx_ref.sum((0, 1, 2))) < 1e-3
    assert relerr(s2, (dx_ref * (x - mean) * invstd).sum((0, 1, 2))) < 1e-3


@pytest.mark.parametrize("case", [(2, 16, 16, 96, 1, 2, None, 0, 1), (1, 33, 33, 1536, 1, 4, None, 0, 1),
                                  (3, 64, 64, 144, 1, 1, None, 0, 2), (1, 64, 64, 96, 1, 12, None, 2, 1),
                                  (2, 17, 19, 40, 1, 3, None, 0, None)])
def test_dwconv_bwd_sums_against_another_tensor(L, case):
    """dl3_dwconv3x3_bwd_sx (round 4): the same gradients bit for bit, the second BatchNorm-backward sum taken against
    x_hat of ANOTHER tensor (the other input of the residual Add whose output gradient the launch completes)"""
    N, H, W, C, stride, rate, pads, impl, act = case
    rng = np.random.default_rng(18)
    Ho, Wo, pt, pl = _dw_geom(H, W, stride, rate, pads)
    x = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    other = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, C)).astype(np.float32)
    g = rng.normal(0, 1, (N, Ho, Wo, C)).astype(np.float32)
    yraw = rng.normal(0, 1, (N, Ho, Wo, C)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, C).astype(np.float32) for _ in range(3)]
    add = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    mean = rng.normal(0, 1, C).astype(np.float32)
    invstd = rng.uniform(0.5, 2, C).astype(np.float32)
    s, t, a = _xform(rng, C, act)
    P = L.dl3_dwconv3x3_partials(N, H, W, C, stride, rate, Ho, Wo, impl)
    head = (ptr(dev(g)), ptr(dev(yraw)), ptr(dev(cA)), ptr(dev(cB)), ptr(dev(cC)), ptr(dev(x)),
            ptr(dev(s)) if s is not None else None, ptr(dev(t)) if t is not None else None, a, ptr(dev(w)))
    tail = (ptr(dev(mean)), ptr(dev(invstd)))
    geom = (N, H, W, C, stride, rate, pt, pl, Ho, Wo, impl)
    dx0, dp0, wp0 = empty(N, H, W, C), empty(P, C, 2), empty(P, 9, C)
    dx1, dp1, wp1 = empty(N, H, W, C), empty(P, C, 2), empty(P, 9, C)
    addd = dev(add)
    call("dl3_dwconv3x3_bwd", *head, ptr(dx0), ptr(addd), *tail, ptr(dp0), ptr(wp0), *geom)
    call("dl3_dwconv3x3_bwd_sx", *head, ptr(dx1), ptr(addd), ptr(dev(other)), *tail, ptr(dp1), ptr(wp1), *geom)
    assert np.array_equal(host(dx0), host(dx1)) and np.array_equal(host(wp0), host(wp1))
    d0, d1 = host(dp0).reshape(P, C, 2), host(dp1).reshape(P, C, 2)
    assert np.array_equal(d0[:, :, 0], d1[:, :, 0])                      # sum(dx): the same
    dxv = host(dx1).astype(np.float64)
    s2 = d1[:, :, 1].astype(np.float64).sum(0)
    assert relerr(s2, (dxv * (other - mean) * invstd).sum((0, 1, 2))) < 1e-3
    # the gather kernels do not take it: refused, not ignored
    if stride == 1:
        assert L.dl3_dwconv3x3_bwd_sx(*head, ptr(dx1), ptr(addd), ptr(dev(other)), *tail, ptr(dp1), ptr(wp1),
                                      N, H, W, C, stride, rate, pt, pl, Ho, Wo, 1, stream()) == -4


@pytest.mark.parametrize("ppb", [2, 3, 5])
def test_dwconv_phases_per_workgroup(L, ppb, monkeypatch):
    """march kernels with a forced number of row phases per workgroup (uneven last group, rate 5 on 16 rows)"""
    monkeypatch.setenv("DL3_DW_PPB", str(ppb))
    case = (2, 16, 20, 40, 1, 5, None, 2, 2)
    test_dwconv_fwd(L, case)
    test_dwconv_bwd(L, case)


@pytest.mark.parametrize("rate", [12, 24, 36])
@pytest.mark.parametrize("ppb", [1, 4, 7])
def test_dwconv_aspp_rates_forced_phases(L, rate, ppb, monkeypatch):
    """the large-rate multi-phase path (DL3_DW_PPB) at the ASPP rates themselves, on a 64x64 map with live side taps"""
    monkeypatch.setenv("DL3_DW_PPB", str(ppb))
    case = (1, 64, 64, 96, 1, rate, None, 2, 1)
    test_dwconv_fwd(L, case)
    test_dwconv_bwd(L, case)


def test_dwconv_bwd_plain_operand(L):
    """cA == NULL (dY = g), no add, no stats, no dx"""
    N, H, W, C = 2, 16, 16, 32
    rng = np.random.default_rng(2)
    x = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    w = rng.normal(0, 0.3, (3, 3, C)).astype(np.float32)
    g = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    tape = O.Tape()
    xv, wv = x.astype(np.float64), w.astype(np.float64)
    yref = O.depthwise3x3(xv, wv, 1, 2, 2, 2, H, W, tape=tape)
    grads = tape.backward(yref, g.astype(np.float64))
    P = L.dl3_dwconv3x3_partials(N, H, W, C, 1, 2, H, W, 0)
    dx, wpart = empty(N, H, W, C), empty(P, 9, C)
    call("dl3_dwconv3x3_bwd", ptr(dev(g)), None, None, None, None, ptr(dev(x)), None, None, 0, ptr(dev(w)), ptr(dx),
         None, None, None, None, ptr(wpart), N, H, W, C, 1, 2, 2, 2, H, W, 0)
    assert relerr(host(dx), grads[id(xv)]) < TOL
    assert relerr(host(wpart).reshape(P, 3, 3, C).sum(0), grads[id(wv)]) < 1e-3


PW_CASES = [
    # M, K, N, ldx_extra, ldy_extra, bias, act
    (256, 16, 96, 0, 0, False, 2),
    (512, 96, 24, 0, 0, False, 2),
    (300, 32, 21, 0, 0, True, None),     # N not a multiple of 4 -> scalar weight loads (16-byte activation loads), M tail
    (4100, 256, 21, 0, 0, True, 1),      # the logits layer (deeplabv3p.py:438): K = 256, N = classes
    (300, 30, 21, 0, 0, True, None),     # neither operand a multiple of 4: scalar loads on both
    (1024, 160, 960, 0, 0, False, None),
    (384, 960, 160, 0, 0, False, 2),
    (130, 24, 144, 0, 0, False, 2),
    (64, 320, 256, 0, 256, False, None),  # writes a channel slice of a concat buffer
    (200, 512, 256, 0, 0, False, 1),
    (4, 320, 256, 0, 0, False, None),    # image-pooling branch: M = batch
    (256, 256, 1344, 0, 0, True, None),  # Subpixel head
    (256, 64, 384, 64, 0, False, 2),     # reads a channel slice
    # shapes only Xception has (deeplabv3p.py:272-313,:414-429): 728 = 22.75 x 32, 1536, 2048, decoder 304 / 48
    (520, 728, 728, 0, 0, False, 1),
    (300, 1024, 1536, 0, 0, False, 1),
    (260, 1536, 2048, 0, 0, False, 1),
    (200, 2048, 256, 0, 1024, False, 1),  # aspp pointwise into a slice of the 1280-wide concat buffer
    (384, 304, 256, 0, 0, False, 1),     # decoder_conv0_pointwise
    (384, 256, 48, 0, 256, False, None),  # feature_projection0 into the [x, dec_skip1] concat slice
    (128, 256, 728, 0, 0, False, 1),
    # small batches (round 5, pw_ksplit32_kernel: 1 024 - 16 384 rows, a 96- or 160-wide output, reduction >= 192 split over the waves)
    (2048, 960, 160, 0, 0, False, 2),
    (1100, 576, 96, 0, 0, True, 1),      # ragged last 32-row tile, bias
    (8192, 384, 96, 32, 0, False, 2),    # reads a channel slice
    (4096, 256, 144, 0, 16, False, 1),   # 144 columns in five blocks, writes a channel slice
    (1056, 196, 132, 0, 0, True, None),  # reduction not a multiple of the K-tile, ragged column block
    (2048, 384, 64, 0, 0, False, 2),     # 64-wide output: two column blocks per wave
]


@pytest.mark.parametrize("case", PW_CASES)
def test_pwconv_fwd(L, case):
    M, K, N, lxe, lye, bias, act = case
    rng = np.random.default_rng(3)
    ldx, ldy = K + lxe, N + lye
    xfull = rng.normal(0, 1, (M, ldx)).astype(np.float32)
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    b = rng.normal(0, 1, N).astype(np.float32) if bias else None
    s, t, a = _xform(rng, K, act)
    xoff = lxe
    x = xfull[:, xoff:xoff + K].astype(np.float64)
    xin = x if s is None else np_act(s * x + t, a)
    ref = xin @ w.astype(np.float64) + (b if bias else 0)
    P = L.dl3_pwconv_partials(M, K, N)
    yfull = torch.zeros(M, ldy, dtype=torch.float32, device="cuda")
    part = empty(P, N, 2)
    xd = dev(xfull)
    call("dl3_pwconv_fwd", ptr(xd, xoff), ldx, ptr(dev(s)) if s is not None else None,
         ptr(dev(t)) if t is not None else None, a, ptr(dev(w)), ptr(dev(b)) if bias else None, ptr(yfull, lye), ldy,
         M, K, N, ptr(part))
    y = host(yfull)
    assert relerr(y[:, lye:], ref) < TOL
    if lye:
        assert np.all(y[:, :lye] == 0)
    s1, s2 = fold_partials(part, P, N)
    assert relerr(s1, ref.sum(0)) < 1e-3
    assert relerr(s2, (ref ** 2).sum(0)) < 1e-3


@pytest.mark.parametrize("cfg", range(7))
def test_pwconv_every_tile_configuration(L, cfg, monkeypatch):
    """the automatic choice depends on the shape (32-row tiles for small M, 128/256-row tiles otherwise): force every
    tile configuration of the stream kernel over the same forward and bwd-data cases"""
    monkeypatch.setenv("DL3_GEMM_CFG", str(cfg))
    test_pwconv_fwd(L, (1000, 160, 960, 0, 0, False, 2))
    test_pwconv_fwd(L, (96, 64, 384, 64, 0, True, None))
    test_pwconv_fwd(L, (300, 24, 144, 0, 0, False, 2))     # two K-tiles, the second one ragged
    test_pwconv_fwd(L, (200, 16, 96, 0, 0, False, None))   # a single K-tile
    test_pwconv_fwd(L, (520, 960, 160, 0, 0, False, 2))    # sixty K-tiles
    test_pwconv_bwd_data(L, (520, 160, 960, 2, True, 1, True))
    test_pwconv_bwd_data(L, (256, 320, 256, None, True, 2, True))
    test_pwconv_bwd_data(L, (200, 144, 24, 2, True, 1, True))
    test_pwconv_bwd_data(L, (520, 960, 160, 2, False, 1, True))  # single-tensor operand, mask + residual gradient


@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_pwconv_split_math(L, cfg, monkeypatch):
    """DL3_GEMM_MATH=split: fp32 operands cut exactly into three bf16 pieces, six of the nine piece products on
    v_mfma_f32_32x32x16_bf16 (csrc/pwgemm.hip split3) — same cases, same tolerances as the f32 MFMA path"""
    monkeypatch.setenv("DL3_GEMM_MATH", "split")
    if cfg >= 0:
        monkeypatch.setenv("DL3_GEMM_CFG", str(cfg))
    test_pwconv_fwd(L, (1000, 160, 960, 0, 0, False, 2))
    test_pwconv_fwd(L, (96, 64, 384, 64, 0, True, None))
    test_pwconv_fwd(L, (300, 24, 144, 0, 0, False, 2))
    test_pwconv_bwd_data(L, (520, 160, 960, 2, True, 1, True))
    test_pwconv_bwd_data(L, (256, 320, 256, None, True, 2, True))
    test_pwconv_bwd_data(L, (300, 256, 21, None, False, 0, False))
    if cfg < 0:
        test_pwconv_bwd_data(L, (65536 + 200, 960, 160, 2, True, 0, True))
        test_pwconv_bwd_data(L, (520, 728, 728, 1, True, 1, True))


@pytest.mark.parametrize("cfg", [-1] + list(range(10)))
def test_pwconv_bwd_weight_split_math(L, cfg, monkeypatch):
    """the weight-gradient kernel in split math: every tile configuration, same cases and tolerances as the f32 MFMA"""
    monkeypatch.setenv("DL3_GEMM_MATH", "split")
    if cfg >= 0:
        monkeypatch.setenv("DL3_WGRAD_CFG", str(cfg))
    for case in BW_CASES[:4] + BW_CASES[5:8]:
        test_pwconv_bwd_weight(L, case)


def test_gemm_math_api(L, monkeypatch):
    """dl3_set_gemm_math overrides the environment; DL3_MATH_ENV hands the choice back to it"""
    from dl3_amd import capi
    monkeypatch.delenv("DL3_GEMM_MATH", raising=False)
    assert capi.get_gemm_math() == "f32"
    capi.set_gemm_math("split")
    try:
        assert capi.get_gemm_math() == "split"
        test_pwconv_fwd(L, (66000, 160, 960, 0, 0, False, 2))
        monkeypatch.setenv("DL3_GEMM_MATH", "f32")
        assert capi.get_gemm_math() == "split"
    finally:
        capi.set_gemm_math(None)
    assert capi.get_gemm_math() == "f32"
    monkeypatch.setenv("DL3_GEMM_MATH", "split")
    assert capi.get_gemm_math() == "split"
    assert L.dl3_set_gemm_math(7) != 0


def test_split_math_error(L, monkeypatch):
    """error of the split-math GEMM against float64, next to the f32 MFMA's: both must be fp32-roundoff class"""
    M, K, N = 65536, 960, 160   # enough row tiles for the 128-row configurations (the 32-row ones keep the f32 MFMA)
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 1, (K, N)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)   # sum_k |a||b| per output element
    errs = {}
    for mode in ("f32", "split"):
        monkeypatch.setenv("DL3_GEMM_MATH", mode)
        y = empty(M, N)
        call("dl3_pwconv_fwd", ptr(dev(x)), K, None, None, 0, ptr(dev(w)), None, ptr(y), N, M, K, N, None)
        errs[mode] = float((np.abs(host(y) - ref) / scale).max())
    print("max |err| / sum|a||b|: f32 MFMA %.3g, split (3 x bf16) %.3g" % (errs["f32"], errs["split"]))
    assert errs["f32"] < 6e-7 and errs["split"] < 6e-7
    assert errs["split"] < 3 * errs["f32"]


BD_CASES = [
    # M, K, N, act, two-tensor, add mode (0 none, 1 tensor, 2 broadcast), stats
    (256, 16, 96, 2, True, 1, True),
    (300, 256, 21, None, False, 0, False),   # logits conv: plain dY, K of the GEMM = 21
    (384, 160, 960, None, True, 1, True),
    (512, 960, 160, 2, True, 0, True),
    (256, 320, 256, None, True, 2, True),    # + broadcast per-image addend (global-pool branch)
    (130, 144, 24, 2, True, 0, True),
    (256, 512, 256, 1, True, 0, True),
    (520, 728, 728, 1, True, 1, True),       # Xception middle flow (+ residual gradient)
    (260, 1536, 2048, 1, True, 0, True),
    (300, 1024, 1536, 1, True, 0, True),
    (200, 2048, 256, 1, True, 0, True),      # aspp pointwise: dX is 2048 wide
    (384, 304, 256, 1, True, 0, True),
    (384, 256, 48, 1, True, 0, False),
    (2048, 160, 960, 2, True, 1, True),      # small batches (pw_ksplit32_kernel): two-tensor operand + residual gradient + sums
    (1100, 96, 576, 1, True, 0, True),       # ragged last row tile, three column blocks
    (4096, 160, 320, None, True, 2, True),   # per-image addend
    (2048, 64, 384, None, True, 1, True),    # 64-wide gradient
    # short reduction into a wide output, >= 512 row tiles: the variant that prefetches the mask operand before the
    # main loop (128x96 tiles; the ragged last row tile takes the generic epilogue)
    (65536 + 200, 960, 160, 2, True, 0, True),
    (65536, 576, 96, 2, False, 0, True),
    (66000, 192, 32, 1, True, 0, False),
    (65536, 384, 64, None, True, 0, True),
]


@pytest.mark.parametrize("case", BD_CASES)
def test_pwconv_bwd_data(L, case):
    M, K, N, act, two, addmode, stats = case
    rng = np.random.default_rng(4)
    g = rng.normal(0, 1, (M, N)).astype(np.float32)
    yraw = rng.normal(0, 1, (M, N)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, N).astype(np.float32) for _ in range(3)]
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    s, t, a = _xform(rng, K, act)
    mean = rng.normal(0, 1, K).astype(np.float32)
    invstd = rng.uniform(0.5, 2, K).astype(np.float32)
    dY = (cA * g.astype(np.float64) + cB * yraw + cC) if two else g.astype(np.float64)
    ref = dY @ w.astype(np.float64).T
    if s is not None:
        ref = ref * np_mask(s * x.astype(np.float64) + t, a)
    add_div, add_scale, add = 1, 1.0, None
    if addmode == 1:
        add = rng.normal(0, 1, (M, K)).astype(np.float32)
        ref = ref + add
    elif addmode == 2:
        add_div, add_scale = 64, 0.25
        add = rng.normal(0, 1, (M // add_div, K)).astype(np.float32)
        ref = ref + add_scale * np.repeat(add, add_div, axis=0)
    wT = empty(N, K)
    call("dl3_transpose", ptr(dev(w)), ptr(wT), K, N)
    assert np.array_equal(host(wT), w.T)
    P = L.dl3_pwconv_partials(M, N, K)
    dx, dpart = empty(M, K), empty(P, K, 2)
    need_x = s is not None or stats
    call("dl3_pwconv_bwd_data", ptr(dev(g)), N, ptr(dev(yraw)) if two else None, N, ptr(dev(cA)) if two else None,
         ptr(dev(cB)) if two else None, ptr(dev(cC)) if two else None, ptr(wT), ptr(dx), K,
         ptr(dev(x)) if need_x else None, K, ptr(dev(s)) if s is not None else None,
         ptr(dev(t)) if t is not None else None, a, ptr(dev(add)) if add is not None else None, K, add_div, add_scale,
         ptr(dev(mean)) if stats else None, ptr(dev(invstd)) if stats else None, ptr(dpart) if stats else None, M, K, N)
    assert relerr(host(dx), ref) < TOL
    if stats:
        s1, s2 = fold_partials(dpart, P, K)
        assert relerr(s1, ref.sum(0)) < 1e-3
        assert relerr(s2, (ref * (x - mean) * invstd).sum(0)) < 1e-3


MSK_CASES = [
    # single-tensor masked bwd-data (dY materialised by dl3_pwconv_bwd_weight_dy: g is the only operand tensor, cA = NULL)
    # with activation mask, BatchNorm-backward sums and every kind of addend.  M, K, N, act, add mode
    (65536, 960, 160, 2, 0),          # project conv, 128x160 tiles (128x96 prefetching tiles unless DL3_GEMM_PRE=0)
    (65536 + 200, 160, 960, None, 1),  # expand conv + residual gradient, ragged last row tile
    (65536, 576, 96, 2, 0),
    (8192, 384, 64, None, 1),         # few row tiles: 32-row configurations
    (4096, 320, 256, None, 2),        # per-image addend (64 rows per image)
    (1000, 144, 24, 2, 0),            # ragged in both directions
    (32768, 728, 728, 1, 1),          # Xception middle flow
    (8192, 160, 960, 2, 0),           # small batches (pw_ksplit32_kernel): B=2's 64x64 maps
    (4128, 96, 576, None, 1),
]


@pytest.mark.parametrize("pre", ["1", "0"])
@pytest.mark.parametrize("case", MSK_CASES)
def test_pwconv_bwd_data_single_tensor_masked(L, case, pre, monkeypatch):
    monkeypatch.setenv("DL3_GEMM_PRE", pre)
    M, K, N, act, addmode = case
    test_pwconv_bwd_data(L, (M, K, N, act, False, addmode, True))


BW_CASES = [
    # M, K, N, act, two-tensor, dbias
    (1024, 16, 96, 2, True, False),
    (2048, 160, 960, None, True, False),
    (2048, 960, 160, 2, True, False),
    (1000, 256, 21, 1, False, True),
    (4, 320, 256, None, True, False),
    (4096, 96, 576, 2, True, False),
    (512, 512, 256, 1, True, False),
    (700, 24, 144, None, True, False),
    (1040, 728, 728, 1, True, False),        # Xception-only shapes
    (600, 1536, 2048, 1, True, False),
    (600, 2048, 256, 1, True, False),
    (768, 304, 256, 1, True, False),
    (768, 256, 48, 1, True, False),
    (520, 1024, 1536, 1, True, False),
]


@pytest.mark.parametrize("case", BW_CASES)
def test_pwconv_bwd_weight(L, case):
    M, K, N, act, two, dbias = case
    rng = np.random.default_rng(5)
    g = rng.normal(0, 1, (M, N)).astype(np.float32)
    yraw = rng.normal(0, 1, (M, N)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, N).astype(np.float32) for _ in range(3)]
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    s, t, a = _xform(rng, K, act)
    xin = x.astype(np.float64) if s is None else np_act(s * x.astype(np.float64) + t, a)
    dY = (cA * g.astype(np.float64) + cB * yraw + cC) if two else g.astype(np.float64)
    ref = xin.T @ dY
    nbytes = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device="cuda")
    dw, db = empty(K, N), empty(N)
    call("dl3_pwconv_bwd_weight", ptr(dev(x)), K, ptr(dev(s)) if s is not None else None,
         ptr(dev(t)) if t is not None else None, a, ptr(dev(g)), N, ptr(dev(yraw)) if two else None, N,
         ptr(dev(cA)) if two else None, ptr(dev(cB)) if two else None, ptr(dev(cC)) if two else None, ptr(dw),
         ptr(db) if dbias else None, M, K, N, ptr(ws), nbytes)
    assert relerr(host(dw), ref) < TOL
    if dbias:
        assert relerr(host(db), dY.sum(0)) < TOL


@pytest.mark.parametrize("case", [c for c in BW_CASES if c[4] and not c[5]] + [(130, 24, 144, 2, True, False)])
def test_pwconv_bwd_weight_writes_dy(L, case):
    """dl3_pwconv_bwd_weight_dy: the same weight gradient bit for bit, plus dY = cA*g + cB*y + cC written once"""
    M, K, N, act, two, dbias = case
    rng = np.random.default_rng(15)
    g = rng.normal(0, 1, (M, N)).astype(np.float32)
    yraw = rng.normal(0, 1, (M, N)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, N).astype(np.float32) for _ in range(3)]
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    s, t, a = _xform(rng, K, act)
    nbytes = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device="cuda")
    args = (ptr(dev(x)), K, ptr(dev(s)) if s is not None else None, ptr(dev(t)) if t is not None else None, a,
            ptr(dev(g)), N, ptr(dev(yraw)), N, ptr(dev(cA)), ptr(dev(cB)), ptr(dev(cC)))
    dw0, dw1 = empty(K, N), empty(K, N)
    ldd = N + 8
    dy = empty(M, ldd)
    call("dl3_pwconv_bwd_weight", *args, ptr(dw0), None, M, K, N, ptr(ws), nbytes)
    call("dl3_pwconv_bwd_weight_dy", *args, ptr(dw1), None, M, K, N, ptr(ws), nbytes, ptr(dy), ldd)
    assert np.array_equal(host(dw0), host(dw1))
    got = host(dy)
    # fp32 fma of the kernel against the same expression in float64
    ref = cA.astype(np.float64) * g + cB.astype(np.float64) * yraw + cC
    assert relerr(got[:, :N], ref) < 1e-6
    assert np.isnan(got[:, N:]).all()   # nothing written beyond the N columns


@pytest.mark.parametrize("cfg", range(10))
def test_pwconv_bwd_weight_dy_every_tile_configuration(L, cfg, monkeypatch):
    """dl3_pwconv_bwd_weight_dy under every tile configuration: K and N ragged against every tile size (1 - 7 workgroups share
    a row slab and take turns with the dY stores), M not a multiple of the 16-row stage, fewer rows than one stage"""
    monkeypatch.setenv("DL3_WGRAD_CFG", str(cfg))
    for case in ((1003, 200, 328, 2, True, False), (2570, 960, 160, 1, True, False), (333, 96, 576, None, True, False),
                 (40, 160, 960, 2, True, False)):
        test_pwconv_bwd_weight_writes_dy(L, case)
        test_pwconv_bwd_weight(L, case)


@pytest.mark.parametrize("case", [(65536 + 8, 160, 960, 2, True, False), (65536, 960, 160, 2, True, False),
                                  (65536 + 24, 96, 576, 2, True, False), (65536, 576, 160, 1, True, False),
                                  (65536, 736, 736, 1, True, False)])
def test_pwconv_bwd_weight_dy_benchmark_routes(L, case):
    """... and at the row counts of the benchmarked plans (M >= 32 768 picks the tiles by the measured shortcuts): the
    160-wide tiles with 1, 5 and 8 workgroups per row slab, the half-empty last tile row / column of K = 960 / 576 and
    N = 960 / 576, Xception's 736 x 736"""
    test_pwconv_bwd_weight_writes_dy(L, case)


@pytest.mark.parametrize("case", [(40000, 160, 960, 2, False, False), (40000 + 9, 96, 576, None, False, False),
                                  (33000, 160, 328, 1, True, False), (32768 + 3, 96, 576, 2, True, False)])
def test_pwconv_bwd_weight_one_tile_row(L, case):
    """round 6 (pw_wgrad_row_kernel): K fits one tile row — plain dY (no BatchNorm behind the convolution) and the two-tensor
    operand, without the dY store; N ragged against the 128-wide tiles (idle waves), M not a multiple of the 16-row stage"""
    assert L.dl3_pwconv_route(2, case[0], case[1], case[2]) == 4
    test_pwconv_bwd_weight(L, case)


@pytest.mark.parametrize("case", [(266000 + 17, 160, 960, 0, 0, False, 2), (140000, 96, 576, 0, 0, True, 1),
                                  (133000, 160, 320, 0, 64, False, None), (131072 + 5, 160, 1000, 0, 0, False, 2),
                                  (131072 + 33, 64, 384, 64, 0, False, 2), (140000, 64, 200, 0, 0, False, None),
                                  (98304 + 7, 160, 960, 0, 0, False, 2), (100000, 96, 576, 0, 0, True, 1)])
def test_pwconv_fwd_weight_stationary_short_reductions(L, case):
    """round 6 (pw_ws2_kernel, forward): a reduction of 160 / 96 into an output at least twice as wide from 131 072 rows — the
    whole weight slice of a column tile resident in LDS, eight waves walking 32-row tiles without a barrier: ragged last row
    tile, a last column tile with 8 of its 160 columns (N = 1 000), bias, an output that is a channel slice"""
    assert L.dl3_pwconv_partials(case[0], case[1], case[2]) >= 1 and L.dl3_pwconv_route(0, case[0], case[1], case[2]) == 2
    test_pwconv_fwd(L, case)


@pytest.mark.parametrize("case", [(262144 + 19, 960, 160, 2, False, 0, True), (131072, 576, 96, 2, False, 0, True),
                                  (140000, 1000, 160, 1, False, 0, True), (133000, 384, 96, None, False, 0, True),
                                  (131072 + 7, 320, 160, 2, False, 0, False), (131072 + 40, 384, 64, None, False, 0, True),
                                  (150000, 200, 64, 2, False, 0, True),
                                  (65536 + 40, 960, 160, 2, False, 0, True), (70000, 576, 96, 2, False, 0, True),   # B=16's row count
                                  # the expand convolutions 64 -> 384: two-tensor operand over a reduction of 384, residual gradient
                                  (131072 + 50, 64, 384, None, True, 1, True), (140000, 64, 384, None, True, 0, True),
                                  (131072, 64, 384, 1, True, 1, False)])
def test_pwconv_bwd_data_weight_stationary_short_reductions(L, case):
    """... and its bwd-data instantiation (single-tensor dY): activation mask from the forward input requested in the row-piece
    layout the output leaves in, BatchNorm-backward sums kept per lane for four columns and folded over the row lanes at the
    end; ragged last row tile, ragged last column tile (K = 1 000), no mask, no sums"""
    test_pwconv_bwd_data(L, case)


@pytest.mark.parametrize("case", [(32768, 256, 21, 0, 0, True, None), (40000 + 13, 256, 21, 0, 0, True, 1),
                                  (8192 + 31, 256, 2, 0, 0, False, 2), (20000, 256, 32, 0, 0, True, None),
                                  (9000, 256, 19, 64, 0, True, 1), (16384, 256, 24, 0, 0, False, 1)])
def test_pwconv_fwd_logits_packed_rows(L, case):
    """round 6 (pw_ws2_kernel FLAT): the logits layer (deeplabv3p.py:438) — N = classes <= 32 columns, rows of N floats back to
    back: a 32-row tile leaves as one contiguous run of 16-byte pieces.  Odd N (21, 19), binary segmentation (2), a full
    column block (32), N a multiple of 4 (24), ragged last row tile, an input that is a channel slice, BatchNorm sums"""
    M, K, N = case[:3]
    assert L.dl3_pwconv_route(0, M, K, N) == 5 and L.dl3_pwconv_route(0, 4096, K, N) != 5
    test_pwconv_fwd(L, case)


@pytest.mark.parametrize("case", [(32768, 256, 21), (40000 + 13, 256, 21), (16384 + 5, 256, 2), (20000, 256, 32), (19000, 256, 19),
                                  (16384, 256, 24), (16400, 256, 1)])
def test_pwconv_bwd_data_logits_narrow_reduction(L, case):
    """round 6 (pw_narrowk_kernel): bwd-data of the logits layer — dX[M, 256] over a reduction of `classes` floats per row, the
    gradient rows taken as one contiguous run per 32-row tile; odd / even / single-class reductions, ragged last row tile"""
    M, K, N = case
    assert L.dl3_pwconv_route(3, M, K, N) == 5 and L.dl3_pwconv_route(3, 4096, K, N) != 5
    test_pwconv_bwd_data(L, (M, K, N, None, False, 0, False))


@pytest.mark.parametrize("case", [(131072, 256, 21, None, False, True), (140000 + 13, 256, 21, 1, False, True),
                                  (131072 + 5, 256, 2, 2, False, True), (132000 + 2, 256, 32, None, False, False),
                                  (133000 + 1, 256, 19, 1, False, True), (300000 + 3, 256, 21, None, False, True)])
def test_pwconv_bwd_weight_logits_narrow_output(L, case):
    """round 6 (pw_wgrad_narrow_kernel): the logits layer's weight and bias gradient — every wave holds the whole 256 x N
    gradient, 4-row chunks through a register ring, the waves of a workgroup meet in LDS in wave order; M not a multiple of the
    4-row chunk, with / without an input transform, without the bias gradient"""
    M, K, N = case[:3]
    assert L.dl3_pwconv_route(4, M, K, N) == 5 and L.dl3_pwconv_route(4, 4096, K, N) != 5
    test_pwconv_bwd_weight(L, case)


def test_logits_layer_routes_at_the_benchmarked_shape(L):
    """the three launches of the logits layer take their narrow kernels at the benchmarked row count (DL3_NARROW, read once per
    process, is the A/B switch: profiles/r06_ab_calls.txt calls 24 / 25), and the forward's partial rows are sized for it"""
    M = 128 * 128 * 128
    assert [L.dl3_pwconv_route(d, M, 256, 21) for d in (0, 3, 4)] == [5, 5, 5]
    assert L.dl3_pwconv_partials(M, 256, 21) >= 256


WIDE_CASES = [(1300, 736, 736), (52480 + 37, 736, 736), (700, 64, 416), (256, 2048, 256)]


@pytest.mark.parametrize("M,K,N", WIDE_CASES)
def test_pwconv_long_reduction_few_row_tiles(L, M, K, N):
    """round 6: launches with a long reduction and at most 1 024 row tiles (Xception at B = 16 / 32: 736 -> 736 at 65 536 rows)
    leave the persistent 512-workgroup loop for one or two tiles per workgroup (gemm_grid_y): forward with BatchNorm sums,
    two-tensor and single-tensor bwd-data with mask, addend and sums, a ragged last row tile, N = 4 x 160 + 96"""
    test_pwconv_fwd(L, (M, K, N, 0, 0, False, 1))
    test_pwconv_fwd(L, (M, K, N, 0, 32, True, None))
    if N == K:
        test_pwconv_bwd_data(L, (M, K, N, 1, True, 1, True))
        test_pwconv_bwd_data(L, (M, K, N, 1, False, 1, True))
        test_pwconv_bwd_data(L, (M, K, N, None, False, 0, False))
    else:
        test_pwconv_bwd_data(L, (M, N, K, 1, True, 0, True))    # (the GEMM's output width is the layer's K)


@pytest.mark.parametrize("M,K,N", [(32768 + 64, 256, 608), (40000, 736, 736), (32768, 512, 864)])
def test_pwconv_column_split_23_blocks(L, M, K, N):
    """round 6 (run_gemm, colsplit_shape): an output width of q x 128 + 96 columns (Xception's 728 channels stored 736 wide =
    23 blocks of 32 = 2 x 4 + 3 x 5) is issued as two launches over disjoint column ranges — [0, N - 480) on the 128-wide
    tiles, the last 480 columns on the 160-wide ones — that share the operand rows and the partial-sum rows: forward with bias,
    BatchNorm sums and an output that is a channel slice; bwd-data with two-tensor / single-tensor operand, mask, residual
    gradient (per row / per image) and sums"""
    assert L.dl3_pwconv_partials(M, K, N) >= max(L.dl3_pwconv_partials(M, K, N - 480), L.dl3_pwconv_partials(M, K, 480))
    test_pwconv_fwd(L, (M, K, N, 0, 0, True, 1))
    test_pwconv_fwd(L, (M, K, N, 0, 32, False, None))
    test_pwconv_bwd_data(L, (M, N, K, 1, True, 1, True))     # (the GEMM's output width is the layer's K)
    test_pwconv_bwd_data(L, (M, N, K, 2, False, 0, True))
    test_pwconv_bwd_data(L, (M, N, K, None, False, 2, False))


FUSED_CASES = [
    # M, K, N, act, two-tensor dY, residual addend, stats (0 none, 1 on the forward input, 2 on another tensor)
    (4096 + 17, 16, 96, None, True, True, 1),    # block 1 expand: K below one 32-block, 3 column blocks, ragged rows
    (3000, 32, 16, 2, True, False, 1),           # block 0 project: ReLU6 mask from the depthwise output
    (2500, 24, 144, None, True, True, 2),        # 24 -> 144: 5 column blocks, sums against the Add's other input
    (2048, 144, 24, 2, True, False, 1),          # 144 -> 24: 5 row blocks of the weight matrix, two dX blocks on one wave
    (1000, 96, 24, 1, False, False, 0),          # plain dY, ReLU mask, no sums
    (777, 144, 32, 2, True, False, 1),
    (1500, 64, 64, 1, True, True, 2),            # K = 64: the widest input that may bring an addend / foreign sums
    (640, 64, 64, None, True, False, 1),         # 2 x 2 blocks
    (20, 16, 96, None, True, True, 1),           # fewer rows than one stage
    (40000, 128, 32, 2, True, False, 1),         # 4 x 1, more stages than workgroups
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_pwconv_bwd_fused(L, case):
    """dl3_pwconv_bwd_fused == dl3_pwconv_bwd_weight + dl3_pwconv_bwd_data of the same layer (float64 reference): weight
    gradient (slabs folded by the op or left for the caller), masked data gradient + addend, BatchNorm-backward sums"""
    M, K, N, act, two, has_add, stats = case
    assert L.dl3_pwconv_bwd_fused_supported(M, K, N) == (2 if K <= 64 else 1)
    rng = np.random.default_rng(16)
    g = rng.normal(0, 1, (M, N)).astype(np.float32)
    yraw = rng.normal(0, 1, (M, N)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, N).astype(np.float32) for _ in range(3)]
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    other = rng.normal(0, 1, (M, K)).astype(np.float32)
    s, t, a = _xform(rng, K, act)
    mean = rng.normal(0, 1, K).astype(np.float32)
    invstd = rng.uniform(0.5, 2, K).astype(np.float32)
    add = rng.normal(0, 1, (M, K)).astype(np.float32) if has_add else None
    z = x.astype(np.float64) if s is None else (s * x.astype(np.float64) + t)
    dY = (cA * g.astype(np.float64) + cB * yraw + cC) if two else g.astype(np.float64)
    dw_ref = np_act(z, a).T @ dY
    dx_ref = dY @ w.astype(np.float64).T * np_mask(z, a)
    if has_add:
        dx_ref = dx_ref + add
    sx = x if stats == 1 else other
    wT = empty(N, K)
    call("dl3_transpose", ptr(dev(w)), ptr(wT), K, N)
    S = L.dl3_pwconv_bwd_fused_splits(M, K, N)
    nbytes = L.dl3_pwconv_bwd_fused_workspace(M, K, N)
    assert nbytes == S * K * N * 4
    xd = dev(x)
    sxd = xd if stats == 1 else dev(other)   # sums against the forward input itself: the SAME device tensor
    args = lambda dwp, dxp, partp, wsp: (
        ptr(xd), K, ptr(dev(s)) if s is not None else None, ptr(dev(t)) if t is not None else None, a, ptr(dev(g)), N,
        ptr(dev(yraw)) if two else None, N, ptr(dev(cA)) if two else None, ptr(dev(cB)) if two else None,
        ptr(dev(cC)) if two else None, ptr(wT), dwp, dxp, K, ptr(dev(add)) if has_add else None, K,
        ptr(sxd) if stats else None, K, ptr(dev(mean)) if stats else None, ptr(dev(invstd)) if stats else None, partp,
        M, K, N, wsp, nbytes)
    ws = empty(S, K, N)
    dw, dx, part = empty(K, N), empty(M, K), empty(S, K, 2)
    call("dl3_pwconv_bwd_fused", *args(ptr(dw), ptr(dx), ptr(part) if stats else None, ptr(ws)))
    assert relerr(host(dx), dx_ref) < TOL
    assert relerr(host(dw), dw_ref) < TOL
    if stats:
        s1, s2 = fold_partials(part, S, K)
        assert relerr(s1, dx_ref.sum(0)) < 1e-3
        assert relerr(s2, (dx_ref * (sx - mean) * invstd).sum(0)) < 1e-3
    # dw == NULL: the slabs stay in the workspace; folded by the caller they are the same gradient bit for bit
    ws2, dx2 = empty(S, K, N), empty(M, K)
    call("dl3_pwconv_bwd_fused", *args(None, ptr(dx2), None if not stats else ptr(empty(S, K, 2)), ptr(ws2)))
    dw2 = empty(K, N)
    call("dl3_reduce_partials", ptr(ws2), S, K * N, ptr(dw2))
    assert np.array_equal(host(dw2), host(dw)) and np.array_equal(host(dx2), host(dx))
    assert L.dl3_pwconv_bwd_fused_supported(M, 192, 64) == 0 and L.dl3_pwconv_bwd_fused_supported(M, 30, 32) == 0
    if K > 64:   # a residual addend next to a wide input is refused, not mishandled
        assert L.dl3_pwconv_bwd_fused(*args(ptr(dw), ptr(dx), None, ptr(ws))[:16], ptr(dx2), K,
                                      *args(ptr(dw), ptr(dx), None, ptr(ws))[18:], stream()) == -4



# ---------------------------------------------------------------------------------------------------------------
# round 5: the weight-stationary streaming forward kernel of the HBM-bound layers (csrc/pwgemm.hip pw_fwd_ws_kernel)
WS_CASES = [
    # M, K, N, ldx_extra, ldy_extra, bias, act, stats — every (K, N) the kernel is instantiated for, ragged last tiles
    (32768 + 77, 16, 96, 0, 0, False, 2, True),
    (40000, 32, 16, 0, 0, False, 2, True),
    (32768 + 5, 96, 24, 0, 0, False, 2, True),
    (33000, 24, 144, 0, 0, False, None, True),
    (32800, 144, 24, 8, 0, False, 2, True),      # reads a channel slice
    (36000, 144, 32, 0, 8, True, 2, True),       # writes a channel slice, bias (the centred frozen BatchNorm's -mean)
    (32768 + 31, 32, 192, 0, 0, False, None, False),
    (50000, 192, 32, 0, 0, False, 2, True),
    (32768, 24, 144, 0, 0, False, 2, False),     # whole tiles only, no statistics
]


def _pw_fwd_chunked(L, M, K, N, lxe, lye, bias, act, stats, seed=21, chunk=1 << 16):
    """dl3_pwconv_fwd against float64, the reference evaluated chunk by chunk (M up to millions of rows)"""
    rng = np.random.default_rng(seed)
    ldx, ldy = K + lxe, N + lye
    xfull = rng.normal(0, 1, (M, ldx)).astype(np.float32)
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    b = rng.normal(0, 1, N).astype(np.float32) if bias else None
    s, t, a = _xform(rng, K, act)
    P = L.dl3_pwconv_partials(M, K, N)
    yfull = torch.zeros(M, ldy, dtype=torch.float32, device="cuda")
    part = empty(P, N, 2) if stats else None
    xd = dev(xfull)
    call("dl3_pwconv_fwd", ptr(xd, lxe), ldx, ptr(dev(s)) if s is not None else None,
         ptr(dev(t)) if t is not None else None, a, ptr(dev(w)), ptr(dev(b)) if bias else None, ptr(yfull, lye), ldy,
         M, K, N, ptr(part) if stats else None)
    y = host(yfull)
    w64 = w.astype(np.float64)
    s1r, s2r, worst, scale = np.zeros(N), np.zeros(N), 0.0, 0.0
    for m0 in range(0, M, chunk):
        x = xfull[m0:m0 + chunk, lxe:lxe + K].astype(np.float64)
        xin = x if s is None else np_act(s * x + t, a)
        ref = xin @ w64 + (b if bias else 0)
        worst = max(worst, float(np.abs(y[m0:m0 + chunk, lye:] - ref).max()))
        scale = max(scale, float(np.abs(ref).max()))
        s1r += ref.sum(0)
        s2r += (ref ** 2).sum(0)
    assert worst / scale < TOL
    if lye:
        assert np.all(y[:, :lye] == 0)
    if stats:
        s1, s2 = fold_partials(part, P, N)
        assert relerr(s1, s1r) < 1e-3 and relerr(s2, s2r) < 1e-3
    return yfull, part


@pytest.mark.parametrize("case", WS_CASES)
def test_pwconv_fwd_weight_stationary(L, case):
    M, K, N = case[:3]
    assert L.dl3_pwconv_fwd_impl(M, K, N) == 1 and L.dl3_pwconv_fwd_impl(M // 4, K, N) == 0
    assert L.dl3_pwconv_fwd_impl(M, K + 8, N) == 0   # only the instantiated layer shapes
    _pw_fwd_chunked(L, *case)
    release()


@pytest.mark.parametrize("M,K,N", [(2097152 + 40, 24, 144), (8388608 + 9, 16, 96)])
def test_pwconv_fwd_weight_stationary_full_size(L, M, K, N):
    """the layer sizes of the benchmarked plan (B = 128: 128x128 and 256x256 maps): every row of millions, the statistic
    partial rows bit-identical from launch to launch (static tile -> wave assignment)"""
    assert L.dl3_pwconv_fwd_impl(M, K, N) == 1
    y0, p0 = _pw_fwd_chunked(L, M, K, N, 0, 0, False, 2, True, seed=22)
    y0, p0 = host(y0).copy(), host(p0).copy()
    release()
    y1, p1 = _pw_fwd_chunked(L, M, K, N, 0, 0, False, 2, True, seed=22)
    assert np.array_equal(host(y1), y0) and np.array_equal(host(p1), p0)
    release()


# six-block shapes (round 5) and a six-block expand convolution with addend + foreign sums
FUSED6_CASES = [
    (3000, 32, 192, None, True, True, 2),
    (2100, 32, 192, 2, True, False, 1),
    (4000, 192, 32, 2, True, False, 1),
    (1500, 64, 96, 1, True, True, 2),
    (1500, 96, 64, 2, True, False, 1),
    (36000, 24, 144, None, True, True, 2),       # more stages than workgroups
]


@pytest.mark.parametrize("case", FUSED6_CASES)
def test_pwconv_bwd_fused_six_blocks(L, case):
    test_pwconv_bwd_fused(L, case)


@pytest.mark.parametrize("case", [(32 * 7, 144, 24, 2, True, False, 1), (50, 96, 24, 1, True, False, 1),
                                  (4096 + 31, 100, 32, 2, True, False, 1), (32 * 40, 16, 96, None, True, True, 1)])
def test_pwconv_bwd_fused_stage_structure(L, case):
    """whole stages only (no ragged tail: the peeled last stage never runs), fewer rows than two stages, a partial last
    32-column block of dX (K = 100, K = 16: the lanes beyond K double a valid column — same address, same bits — so that
    every lane issues every store of a whole stage and the wait for the next stage's rows stays a counted one)"""
    test_pwconv_bwd_fused(L, case)


@pytest.mark.parametrize("case", FUSED_CASES[:5])
def test_pwconv_bwd_fused_round4_kernel(L, case, monkeypatch):
    """DL3_FUSED_V=1: the round-4 kernel, kept for same-call A/B measurements"""
    monkeypatch.setenv("DL3_FUSED_V", "1")
    test_pwconv_bwd_fused(L, case)


def test_pwconv_bwd_fused_full_size(L):
    """24 -> 144 at the benchmarked plan's 2 097 152 rows: dX of every row, dW, the BatchNorm-backward sums; twice, bit
    for bit"""
    M, K, N = 2097152 + 50, 24, 144
    rng = np.random.default_rng(23)
    g = rng.normal(0, 1, (M, N)).astype(np.float32)
    yraw = rng.normal(0, 1, (M, N)).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, N).astype(np.float32) for _ in range(3)]
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    add = rng.normal(0, 1, (M, K)).astype(np.float32)
    mean = rng.normal(0, 1, K).astype(np.float32)
    invstd = rng.uniform(0.5, 2, K).astype(np.float32)
    wT = empty(N, K)
    call("dl3_transpose", ptr(dev(w)), ptr(wT), K, N)
    S = L.dl3_pwconv_bwd_fused_splits(M, K, N)
    nbytes = L.dl3_pwconv_bwd_fused_workspace(M, K, N)
    xd, gd, yd, ad = dev(x), dev(g), dev(yraw), dev(add)
    cAd, cBd, cCd, md, isd = dev(cA), dev(cB), dev(cC), dev(mean), dev(invstd)
    outs = []
    for rep in range(2):
        ws, dw, dx, part = empty(S, K, N), empty(K, N), empty(M, K), empty(S, K, 2)
        call("dl3_pwconv_bwd_fused", ptr(xd), K, None, None, 0, ptr(gd), N, ptr(yd), N, ptr(cAd), ptr(cBd), ptr(cCd), ptr(wT),
             ptr(dw), ptr(dx), K, ptr(ad), K, ptr(xd), K, ptr(md), ptr(isd), ptr(part), M, K, N, ptr(ws), nbytes)
        outs.append((host(dx).copy(), host(dw).copy(), host(part).copy()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    dxg, dwg, partg = outs[0]
    w64 = w.astype(np.float64)
    dw_ref, s1r, s2r, worst, scale = np.zeros((K, N)), np.zeros(K), np.zeros(K), 0.0, 0.0
    for m0 in range(0, M, 1 << 16):
        sl = slice(m0, m0 + (1 << 16))
        dY = cA * g[sl].astype(np.float64) + cB * yraw[sl] + cC
        x64 = x[sl].astype(np.float64)
        dw_ref += x64.T @ dY
        dxr = dY @ w64.T + add[sl]
        worst = max(worst, float(np.abs(dxg[sl] - dxr).max()))
        scale = max(scale, float(np.abs(dxr).max()))
        s1r += dxr.sum(0)
        s2r += (dxr * (x64 - mean) * invstd).sum(0)
    assert worst / scale < TOL
    assert relerr(dwg, dw_ref) < TOL
    p = partg.reshape(S, K, 2).astype(np.float64)
    assert relerr(p[:, :, 0].sum(0), s1r) < 1e-3 and relerr(p[:, :, 1].sum(0), s2r) < 1e-3
    release()


# (7 x 510 x 390: 10 877 full 32-pixel tiles + a ragged one — more tiles than the stem kernel has waves, so its software-
# pipelined tile loop runs two and three tiles per wave, with dead taps on every edge of the image)
@pytest.mark.parametrize("shape", [(2, 32, 32, 3, 32, 2), (1, 17, 19, 3, 32, 2), (1, 16, 16, 32, 64, 1), (7, 510, 390, 3, 32, 2)])
def test_conv3x3(L, shape):
    N, H, W, Cin, Cout, stride = shape
    rng = np.random.default_rng(6)
    Ho, pt, _ = O.same_pads(H, 3, stride, 1)
    Wo, pl, _ = O.same_pads(W, 3, stride, 1)
    x = rng.uniform(0, 255, (N, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (3, 3, Cin, Cout)).astype(np.float32)
    s = np.full(Cin, 1 / 127.5, np.float32)
    t = np.full(Cin, -1.0, np.float32)
    xin = s * x.astype(np.float64) + t
    tape = O.Tape()
    wv = w.astype(np.float64)
    ref = O.conv2d(xin, wv, stride, pt, pl, Ho, Wo, tape=tape)
    P = L.dl3_conv3x3_partials(N, Ho, Wo, Cout)
    y, part = empty(N, Ho, Wo, Cout), empty(P, Cout, 2)
    xd, sd, td, wd = dev(x), dev(s), dev(t), dev(w)
    call("dl3_conv3x3_fwd", ptr(xd), ptr(sd), ptr(td), 0, ptr(wd), ptr(y), N, H, W, Cin, Cout, stride, pt, pl, Ho, Wo,
         ptr(part))
    assert relerr(host(y), ref) < TOL
    s1, s2 = fold_partials(part, P, Cout)
    assert relerr(s1, ref.sum((0, 1, 2))) < 1e-3 and relerr(s2, (ref ** 2).sum((0, 1, 2))) < 1e-3
    # weight gradient with a BN-backward operand
    g = rng.normal(0, 1, ref.shape).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, Cout).astype(np.float32) for _ in range(3)]
    yraw = host(y)
    dY = cA * g.astype(np.float64) + cB * yraw + cC
    grads = tape.backward(ref, dY)
    wpart = empty(P, 9 * Cin * Cout)
    call("dl3_conv3x3_bwd_weight", ptr(xd), ptr(sd), ptr(td), 0, ptr(dev(g)), ptr(y), ptr(dev(cA)), ptr(dev(cB)),
         ptr(dev(cC)), ptr(wpart), N, H, W, Cin, Cout, stride, pt, pl, Ho, Wo)
    assert relerr(host(wpart).astype(np.float64).sum(0).reshape(3, 3, Cin, Cout), grads[id(wv)]) < 1e-3
    if Cin % 4 == 0:
        P2 = L.dl3_conv3x3_partials(N, H, W, Cin)
        dx, dpart = empty(N, H, W, Cin), empty(P2, Cin, 2)
        mean = rng.normal(0, 1, Cin).astype(np.float32)
        invstd = rng.uniform(0.5, 2, Cin).astype(np.float32)
        call("dl3_conv3x3_bwd_data", ptr(dev(g)), ptr(y), ptr(dev(cA)), ptr(dev(cB)), ptr(dev(cC)), ptr(wd), ptr(dx),
             ptr(xd), ptr(sd), ptr(td), 1, None, ptr(dev(mean)), ptr(dev(invstd)), ptr(dpart), N, H, W, Cin, Cout,
             stride, pt, pl, Ho, Wo)
        dx_ref = grads[id(xin)] * (xin > 0)
        assert relerr(host(dx), dx_ref) < TOL
        s1, s2 = fold_partials(dpart, P2, Cin)
        assert relerr(s1, dx_ref.sum((0, 1, 2))) < 1e-3
        assert relerr(s2, (dx_ref * (x - mean) * invstd).sum((0, 1, 2))) < 1e-3


@pytest.mark.parametrize("M,K,N,div", [(4 * 96, 40, 48, 96), (3 * 4096, 256, 256, 4096), (2 * 50, 24, 21, 50), (300, 64, 160, 1)])
def test_pwconv_fwd_add(L, M, K, N, div):
    """dl3_pwconv_fwd_add: y = act(s*x+t) . w + bias + add[m // div] — a residual tensor (div = 1) or one row per image
    (the ASPP image-pooling branch as a per-image term): the straight-line epilogue (32-row blocks inside one image),
    the predicated one (div = 50), full and ragged tiles; BatchNorm partial sums include the addend"""
    rng = np.random.default_rng(M + N)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.normal(0, 0.3, K).astype(np.float32)
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    add = rng.normal(0, 1, (M // div, N)).astype(np.float32)
    ref = np_act(x.astype(np.float64) * sc + sh, 2) @ w.astype(np.float64) + bias + np.repeat(add.astype(np.float64), div, axis=0)
    P = L.dl3_pwconv_partials(M, K, N)
    y, part = empty(M, N), empty(P, N, 2)
    call("dl3_pwconv_fwd_add", ptr(dev(x)), K, ptr(dev(sc)), ptr(dev(sh)), 2, ptr(dev(w)), ptr(dev(bias)), ptr(y), N, M, K, N,
         ptr(part), ptr(dev(add)), N, div)
    assert relerr(host(y), ref) < 2e-5
    s1, s2 = fold_partials(part, P, N)
    assert relerr(s1, ref.sum(0)) < 1e-4 and relerr(s2, (ref ** 2).sum(0)) < 1e-4


@pytest.mark.parametrize("M,K,N,ldx_extra,ldy_extra,div", [(1, 2048, 256, 0, 0, 0), (3, 320, 256, 0, 0, 0), (128, 256, 256, 0, 0, 1),
                                                          (256, 2048, 256, 64, 8, 0), (6, 96, 21, 0, 0, 3), (2, 2048, 256, 0, 0, 2)])
def test_pwconv_fwd_few_rows_accumulates_in_double(L, M, K, N, ldx_extra, ldy_extra, div):
    """dl3_pwconv_fwd_rows: a forward 1x1 convolution over a handful of rows — the ASPP image-pooling branch, one row per
    image (deeplabv3p.py:375-382), and its per-image share of concat_projection — accumulated in DOUBLE
    (pw_rows_f64_kernel: its result is added to every pixel of the map, its rounding error does not average out): the
    result is the float64 product rounded ONCE, far inside what a float32 reduction of 2 048 terms can do; with input
    transform, bias, per-image / per-row addend, channel slices on both sides; and the global pool in front of it
    (dl3_gap_fwd) sums in double as well"""
    rng = np.random.default_rng(100 * M + N)
    ldx, ldy = K + ldx_extra, N + ldy_extra
    xf = rng.normal(0, 1, (M, ldx)).astype(np.float32)
    w = rng.normal(0, 0.2, (K, N)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, K).astype(np.float32), rng.normal(0, 0.3, K).astype(np.float32)
    bias = rng.normal(0, 0.5, N).astype(np.float32)
    x = xf[:, ldx_extra:]
    t32 = np_act(x * sc + sh, 1)                                   # the transform itself is float32 arithmetic
    ref = t32.astype(np.float64) @ w.astype(np.float64) + bias
    yfull = torch.zeros(M, ldy, dtype=torch.float32, device="cuda")
    xd = dev(xf)
    if div:
        add = rng.normal(0, 1, (M // div, N)).astype(np.float32)
        ref = ref + np.repeat(add.astype(np.float64), div, axis=0)
        call("dl3_pwconv_fwd_rows", ptr(xd, ldx_extra), ldx, ptr(dev(sc)), ptr(dev(sh)), 1, ptr(dev(w)), ptr(dev(bias)),
             ptr(yfull, ldy_extra), ldy, M, K, N, ptr(dev(add)), N, div)
    else:
        call("dl3_pwconv_fwd_rows", ptr(xd, ldx_extra), ldx, ptr(dev(sc)), ptr(dev(sh)), 1, ptr(dev(w)), ptr(dev(bias)),
             ptr(yfull, ldy_extra), ldy, M, K, N, None, 0, 1)
    y = host(yfull)
    err = np.abs(y[:, ldy_extra:] - ref).max() / np.abs(ref).max()
    assert err < 1.5e-7, err                                        # one float32 rounding of the float64 result
    if ldy_extra:
        assert np.all(y[:, :ldy_extra] == 0)
    # the global pool: 4 096 pixels x channels, float64 sums rounded once
    HW, C, B = 4096, 96, 2
    t = rng.normal(3.0, 1.0, (B, HW, C)).astype(np.float32)
    out = empty(B, C)
    call("dl3_gap_fwd", ptr(dev(t)), C, None, None, 0, ptr(out), B, HW, C, 1.0 / HW)
    want = t.astype(np.float64).mean(1)
    assert np.abs(host(out) - want).max() / np.abs(want).max() < 1.5e-7
    if M == 1:
        # 600 (column block, image) pairs: the 256-thread form of the kernel (few pairs take 1 024 threads); ragged HW and C
        HW, C, B = 1000, 70, 200
        t = rng.normal(3.0, 1.0, (B, HW, C)).astype(np.float32)
        out = empty(B, C)
        call("dl3_gap_fwd", ptr(dev(t)), C, None, None, 0, ptr(out), B, HW, C, 1.0 / HW)
        want = t.astype(np.float64).mean(1)
        assert np.abs(host(out) - want).max() / np.abs(want).max() < 1.5e-7


@pytest.mark.parametrize("M", [2, 37, 4096])
def test_bn_finalize_direct_small_tensors(L, M):
    """dl3_bn_finalize_direct: two-pass double statistics straight from a small tensor (the image-pooling BatchNorm,
    deeplabv3p.py:375-379, sees one value per image: at M = 2 the variance of two nearly equal numbers — the case in
    which sum(y^2)/n - mean^2 of fp32 partial sums loses every digit; measured in round 3: mean^2/var 8.6e6)."""
    rng = np.random.default_rng(M)
    C, ld, c0 = 21, 40, 8
    y = np.zeros((M, ld), np.float32)
    mu = rng.uniform(-50, 50, C)
    spread = rng.uniform(1e-3, 2.0, C)          # some channels: |mean| / spread ~ 5e4
    y[:, c0:c0 + C] = (mu + spread * rng.normal(0, 1, (M, C))).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 1, C).astype(np.float32)
    mm, mv = rng.normal(0, 1, C).astype(np.float32), rng.uniform(0.5, 2, C).astype(np.float32)
    eps, mom = 1e-5, 0.99
    y64 = y[:, c0:c0 + C].astype(np.float64)
    mean, var = y64.mean(0), y64.var(0)
    invstd = 1 / np.sqrt(var + eps)
    unb = (M / (M - 1.0)) * (M / (M - (1 + eps)))
    outs = [empty(C) for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    call("dl3_bn_finalize_direct", ptr(dev(y), c0), ld, M, C, ptr(dev(gamma)), ptr(dev(beta)), eps, mom, unb,
         *[ptr(o) for o in outs], ptr(mmd), ptr(mvd))
    sc, sh, me, isd = [host(o) for o in outs]
    assert np.allclose(isd, invstd, rtol=1e-6) and np.allclose(me, mean, rtol=1e-6)
    assert np.allclose(sc, gamma * invstd, rtol=1e-6) and np.allclose(sh, beta - mean * gamma * invstd, rtol=2e-6, atol=1e-6)
    m32 = float(np.float32(mom))  # the C ABI takes the momentum as a float
    assert np.allclose(host(mmd), m32 * mm + (1 - m32) * mean, rtol=1e-6, atol=1e-6)
    assert np.allclose(host(mvd), m32 * mv + (1 - m32) * var * unb, rtol=1e-6, atol=1e-7)
    # what the partial-sum form makes of the worst channel (float32 sums, folded in double — dl3_bn_finalize's input)
    s1, s2 = y[:, c0:c0 + C].sum(0, dtype=np.float32).astype(np.float64), (y[:, c0:c0 + C] ** 2).sum(0, dtype=np.float32).astype(np.float64)
    var_p = np.maximum(s2 / M - (s1 / M) ** 2, 0)
    print("M=%d: worst relative variance error of the partial-sum form %.2e, of the direct form %.2e" % (
        M, float((np.abs(var_p - var) / var).max()), float((np.abs(1 / isd.astype(np.float64) ** 2 - eps - var) / var).max())))


def test_bn_finalize_and_bwd(L):
    rng = np.random.default_rng(7)
    P, ldc, C, c0, count = 37, 48, 24, 16, 1000.0
    part = rng.normal(0, 1, (P, ldc, 2)).astype(np.float32)
    part[:, :, 1] = np.abs(part[:, :, 1]) * 40 + 30  # sum of squares: keep the variance positive
    gamma, beta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 1, C).astype(np.float32)
    mm, mv = rng.normal(0, 1, C).astype(np.float32), rng.uniform(0.5, 2, C).astype(np.float32)
    eps, mom = 1e-3, 0.99
    s1 = part[:, c0:c0 + C, 0].astype(np.float64).sum(0)
    s2 = part[:, c0:c0 + C, 1].astype(np.float64).sum(0)
    mean = s1 / count
    var = s2 / count - mean ** 2
    invstd = 1 / np.sqrt(var + eps)
    outs = [empty(C) for _ in range(4)]
    mmd, mvd = dev(mm), dev(mv)
    pd = dev(part)
    # var_unbias: Keras 2.2.4 on TF 1.13 = n/(n-1) * n/(n-(1+eps)) (include/dl3.h)
    unb = count / (count - 1) * count / (count - (1 + eps))
    call("dl3_bn_finalize", ptr(pd, 2 * c0), P, ldc, C, count, ptr(dev(gamma)), ptr(dev(beta)), eps, mom, unb,
         *[ptr(o) for o in outs], ptr(mmd), ptr(mvd))
    sc, sh, me, isd = [host(o) for o in outs]
    assert relerr(sc, gamma * invstd) < 1e-5 and relerr(sh, beta - mean * gamma * invstd) < 1e-5
    assert relerr(me, mean) < 1e-5 and relerr(isd, invstd) < 1e-5
    assert relerr(host(mmd), mom * mm + (1 - mom) * mean) < 1e-6
    assert relerr(host(mvd), mom * mv + (1 - mom) * var * unb) < 1e-6
    # a non-trainable layer: same (scale, shift), moving statistics not touched (NULL pointers)
    outs3 = [empty(C) for _ in range(4)]
    call("dl3_bn_finalize", ptr(pd, 2 * c0), P, ldc, C, count, ptr(dev(gamma)), ptr(dev(beta)), eps, mom, unb,
         *[ptr(o) for o in outs3], None, None)
    assert np.array_equal(host(outs3[0]), sc) and np.array_equal(host(outs3[1]), sh)
    # frozen
    outs2 = [empty(C) for _ in range(4)]
    call("dl3_bn_frozen", ptr(dev(gamma)), ptr(dev(beta)), ptr(dev(mm)), ptr(dev(mv)), eps, C, *[ptr(o) for o in outs2])
    assert relerr(host(outs2[0]), gamma / np.sqrt(mv + eps)) < 1e-5
    assert relerr(host(outs2[1]), beta - mm * gamma / np.sqrt(mv + eps)) < 1e-5
    # backward coefficients: dy = gamma*invstd*(g - sum(g)/M - xhat*sum(g*xhat)/M)
    dpart = rng.normal(0, 1, (P, ldc, 2)).astype(np.float32)
    d1 = dpart[:, c0:c0 + C, 0].astype(np.float64).sum(0)
    d2 = dpart[:, c0:c0 + C, 1].astype(np.float64).sum(0)
    co = [empty(C) for _ in range(5)]
    call("dl3_bn_bwd_finalize", ptr(dev(dpart), 2 * c0), P, ldc, C, count, ptr(dev(gamma)), ptr(dev(mean)),
         ptr(dev(invstd)), 1, *[ptr(o) for o in co])
    cA, cB, cC, dg, db = [host(o).astype(np.float64) for o in co]
    assert relerr(dg, d2) < 1e-5 and relerr(db, d1) < 1e-5
    g = rng.normal(0, 1, (50, C))
    y = rng.normal(0, 1, (50, C))
    xhat = (y - mean) * invstd
    want = gamma * invstd * (g - d1 / count - xhat * d2 / count)
    assert relerr(cA * g + cB * y + cC, want) < 1e-4
    call("dl3_bn_bwd_finalize", ptr(dev(dpart), 2 * c0), P, ldc, C, count, ptr(dev(gamma)), ptr(dev(mean)),
         ptr(dev(invstd)), 0, *[ptr(o) for o in co])
    assert relerr(host(co[0]), gamma * invstd) < 1e-5 and np.all(host(co[1]) == 0) and np.all(host(co[2]) == 0)


@pytest.mark.parametrize("P,n", [(5, 100), (64, 333), (700, 77), (119, 4096)])
def test_reduce_partials(L, P, n):
    rng = np.random.default_rng(8)
    part = rng.normal(0, 1, (P, n)).astype(np.float32)
    out = empty(n)
    call("dl3_reduce_partials", ptr(dev(part)), P, n, ptr(out))
    assert relerr(host(out), part.astype(np.float64).sum(0)) < 1e-5


def test_reduce_partials_batched_is_bit_identical(L):
    """dl3_reduce_partials_batched (every weight-gradient slab fold of a backward pass in one launch) == one
    dl3_reduce_partials per fold, bit for bit, for every (rows, columns) class of block shape and ragged column counts"""
    rng = np.random.default_rng(81)
    folds = [(3, 700), (32, 64), (33, 1000), (256, 31), (257, 999), (1365, 9 * 40), (1, 5), (17, 153600)]
    parts = [dev(rng.normal(0, 1, (P, n)).astype(np.float32)) for P, n in folds]
    want = []
    for (P, n), pt_ in zip(folds, parts):
        o = empty(n)
        call("dl3_reduce_partials", ptr(pt_), P, n, ptr(o))
        want.append(host(o).copy())
        assert relerr(want[-1], host(pt_).astype(np.float64).sum(0)) < 1e-5
    outs = [torch.full((n,), float("nan"), device="cuda") for _, n in folds]
    rows, b0 = [], 0
    for (P, n), pt_, o in zip(folds, parts, outs):
        rows.append([pt_.data_ptr(), o.data_ptr(), P, n, b0])
        nb = L.dl3_reduce_partials_blocks(P, n)
        assert nb == -(-n // (64 if P <= 32 else 32 if P <= 256 else 8))
        b0 += nb
    desc = torch.tensor(rows, dtype=torch.int64, device="cuda")
    call("dl3_reduce_partials_batched", desc.data_ptr(), len(rows), b0)
    for o, w in zip(outs, want):
        assert np.array_equal(host(o), w)
    assert L.dl3_reduce_partials_blocks(0, 5) == 0


@pytest.mark.parametrize("case", [(4096, 160, 960, 2, True), (1040, 96, 24, 2, True), (600, 320, 256, 1, False)])
def test_pwconv_bwd_weight_slabs_left_for_the_caller(L, case):
    """dw == NULL: the launch leaves its S = dl3_pwconv_bwd_weight_splits slabs [S][K][N] at the head of the workspace;
    folding them afterwards gives exactly what the folding call writes"""
    M, K, N, act, two = case
    rng = np.random.default_rng(82)
    g, yraw, x = [dev(rng.normal(0, 1, sh).astype(np.float32)) for sh in ((M, N), (M, N), (M, K))]
    cA, cB, cC = [dev(rng.normal(0, 1, N).astype(np.float32)) for _ in range(3)]
    s, t = dev(rng.uniform(0.5, 1.5, K).astype(np.float32)), dev(rng.normal(0, 0.5, K).astype(np.float32))
    nbytes = L.dl3_pwconv_bwd_weight_workspace(M, K, N)
    S = L.dl3_pwconv_bwd_weight_splits(M, K, N, 1 if two else 0)
    assert S >= 1 and S * K * N * 4 <= nbytes
    out = []
    for leave in (False, True):
        ws = torch.full((nbytes // 4 + 4,), float("nan"), dtype=torch.float32, device="cuda")
        dw = empty(K, N)
        call("dl3_pwconv_bwd_weight", ptr(x), K, ptr(s), ptr(t), act, ptr(g), N, ptr(yraw) if two else None, N,
             ptr(cA) if two else None, ptr(cB) if two else None, ptr(cC) if two else None, None if leave else ptr(dw),
             None, M, K, N, ptr(ws), nbytes)
        if leave:
            call("dl3_reduce_partials", ptr(ws), S, K * N, ptr(dw))
        out.append(host(dw).copy())
    assert np.isfinite(out[0]).all() and np.array_equal(out[0], out[1])
    from dl3_amd import capi
    with pytest.raises(capi.DL3Error):   # a bias gradient needs the folding call
        call("dl3_pwconv_bwd_weight", ptr(x), K, ptr(s), ptr(t), act, ptr(g), N, None, N, None, None, None, None,
             ptr(empty(N)), M, K, N, ptr(ws), nbytes)


def test_affine_add_and_dropout(L):
    rng = np.random.default_rng(9)
    M, C = 300, 40
    a, b = rng.normal(0, 1, (M, C)).astype(np.float32), rng.normal(0, 1, (M, C)).astype(np.float32)
    sa, ta = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 1, C).astype(np.float32)
    out = empty(M, C)
    call("dl3_affine_add", ptr(dev(a)), C, ptr(dev(sa)), ptr(dev(ta)), 2, ptr(dev(b)), C, None, None, 0, ptr(out), C, M,
         C, 0.0, 0, None)
    assert relerr(host(out), np_act(sa * a + ta, 2) + b) < 1e-6
    # dropout: deterministic mask of the right density, identical between forward and gradient kernels
    M2, C2 = 4096, 256
    ones = np.ones((M2, C2), np.float32)
    o1, o2 = empty(M2, C2), empty(M2, C2)
    call("dl3_affine_add", ptr(dev(ones)), C2, None, None, 0, None, 0, None, None, 0, ptr(o1), C2, M2, C2, 0.1, 1234, None)
    call("dl3_grad_finish", ptr(dev(ones)), C2, 1, 1.0, ptr(o2), C2, None, 0, None, 0, None, None, 0, None, None, None,
         M2, C2, 0.1, 1234, None)
    h1, h2 = host(o1), host(o2)
    assert np.array_equal(h1, h2)
    keep = (h1 != 0).mean()
    assert abs(keep - 0.9) < 5e-3
    assert np.allclose(h1[h1 != 0], 1 / 0.9)
    assert np.array_equal(h1 != 0, dropout_keep_mask(1234, M2 * C2, 0.1).reshape(M2, C2) != 0)
    # the step number lives in device memory (dl3_counter_add): step 0 == no step pointer, every further step draws a
    # new mask, identically in the forward and the gradient kernel, and the host replica follows
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    masks = []
    for it in range(3):
        call("dl3_affine_add", ptr(dev(ones)), C2, None, None, 0, None, 0, None, None, 0, ptr(o1), C2, M2, C2, 0.1, 1234,
             step.data_ptr())
        call("dl3_grad_finish", ptr(dev(ones)), C2, 1, 1.0, ptr(o2), C2, None, 0, None, 0, None, None, 0, None, None,
             None, M2, C2, 0.1, 1234, step.data_ptr())
        call("dl3_counter_add", step.data_ptr(), 1)
        m1, m2 = host(o1) != 0, host(o2) != 0
        assert np.array_equal(m1, m2)
        assert np.array_equal(m1, dropout_keep_mask(1234, M2 * C2, 0.1, step=it).reshape(M2, C2) != 0)
        masks.append(m1)
    assert int(step.item()) == 3
    assert np.array_equal(masks[0], h1 != 0)
    assert 0.15 < (masks[0] != masks[1]).mean() < 0.21 and 0.15 < (masks[1] != masks[2]).mean() < 0.21  # 2*0.9*0.1


def test_grad_finish_and_gap(L):
    rng = np.random.default_rng(10)
    M, C, HW = 512, 48, 128
    g = rng.normal(0, 1, (M, C)).astype(np.float32)
    x = rng.normal(0, 1, (M, C)).astype(np.float32)
    add = rng.normal(0, 1, (M, C)).astype(np.float32)
    s, t = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 1, C).astype(np.float32)
    mean, invstd = rng.normal(0, 1, C).astype(np.float32), rng.uniform(0.5, 2, C).astype(np.float32)
    P = L.dl3_rows_partials(M)
    out, part = empty(M, C), empty(P, C, 2)
    call("dl3_grad_finish", ptr(dev(g)), C, 1, 1.0, ptr(out), C, ptr(dev(add)), C, ptr(dev(x)), C, ptr(dev(s)),
         ptr(dev(t)), 1, ptr(dev(mean)), ptr(dev(invstd)), ptr(part), M, C, 0.0, 0, None)
    ref = g * np_mask(s * x + t, 1) + add
    assert relerr(host(out), ref) < 1e-6
    s1, s2 = fold_partials(part, P, C)
    assert relerr(s1, ref.astype(np.float64).sum(0)) < 1e-4
    assert relerr(s2, (ref.astype(np.float64) * (x - mean) * invstd).sum(0)) < 1e-4
    # broadcast form (backward of the global average pool) accumulating in place
    gv = rng.normal(0, 1, (M // HW, C)).astype(np.float32)
    acc = dev(add)
    call("dl3_grad_finish", ptr(dev(gv)), C, HW, 1.0 / HW, ptr(acc), C, ptr(acc), C, None, 0, None, None, 0, None, None,
         None, M, C, 0.0, 0, None)
    assert relerr(host(acc), add + np.repeat(gv, HW, axis=0) / HW) < 1e-6
    # global average pool with transform, reading a channel slice
    N = M // HW
    xf = rng.normal(0, 1, (M, C + 16)).astype(np.float32)
    o = empty(N, C)
    call("dl3_gap_fwd", ptr(dev(xf), 16), C + 16, ptr(dev(s)), ptr(dev(t)), 1, ptr(o), N, HW, C, 1.0 / HW)
    ref = np_act(s * xf[:, 16:].astype(np.float64) + t, 1).reshape(N, HW, C).mean(1)
    assert relerr(host(o), ref) < 1e-5


@pytest.mark.parametrize("dims", [(2, 8, 8, 64, 64, 21), (2, 1, 1, 16, 16, 32), (1, 16, 16, 32, 32, 8),
                                  (1, 10, 7, 33, 20, 5)])
def test_resize_bilinear(L, dims):
    N, Hi, Wi, Ho, Wo, C = dims
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, (N, Hi, Wi, C)).astype(np.float32)
    tape = O.Tape()
    ref = O.resize_bilinear_tf1(x, Ho, Wo, tape=tape)
    y = empty(N, Ho, Wo, C)
    call("dl3_resize_bilinear_fwd", ptr(dev(x)), C, None, None, 0, ptr(y), C, N, Hi, Wi, Ho, Wo, C)
    assert relerr(host(y), ref) < 1e-5
    g = rng.normal(0, 1, ref.shape).astype(np.float32)
    dx_ref = tape.backward(ref, g)[id(x)]
    dx = empty(N, Hi, Wi, C)
    call("dl3_resize_bilinear_bwd", ptr(dev(g)), C, ptr(dx), C, N, Hi, Wi, Ho, Wo, C, 0, None, 0)
    assert relerr(host(dx), dx_ref) < 1e-4
    call("dl3_resize_bilinear_bwd", ptr(dev(g)), C, ptr(dx), C, N, Hi, Wi, Ho, Wo, C, 1, None, 0)
    assert relerr(host(dx), 2 * dx_ref) < 1e-4
    nb = L.dl3_resize_bilinear_bwd_workspace(N, Hi, Wi, Ho, Wo, C)  # separable two-pass form
    ws = empty(nb // 4 + 4)
    call("dl3_resize_bilinear_bwd", ptr(dev(g)), C, ptr(dx), C, N, Hi, Wi, Ho, Wo, C, 0, ptr(ws), nb)
    assert relerr(host(dx), dx_ref) < 1e-4


def test_subsample(L):
    rng = np.random.default_rng(15)
    N, H, W, C, s_ = 2, 9, 10, 24, 2
    Ho, Wo = (H - 1) // s_ + 1, (W - 1) // s_ + 1
    x = rng.normal(0, 1, (N, H, W, C)).astype(np.float32)
    sc, sh = rng.uniform(0.5, 1.5, C).astype(np.float32), rng.normal(0, 1, C).astype(np.float32)
    y = empty(N, Ho, Wo, C)
    call("dl3_subsample_fwd", ptr(dev(x)), C, ptr(dev(sc)), ptr(dev(sh)), 1, ptr(y), N, H, W, C, s_, Ho, Wo)
    assert relerr(host(y), np_act(sc * x[:, ::s_, ::s_] + sh, 1)) < 1e-6
    g = rng.normal(0, 1, (N, Ho, Wo, C)).astype(np.float32)
    dx = empty(N, H, W, C)
    call("dl3_subsample_bwd", ptr(dev(g)), ptr(dx), N, H, W, C, s_, Ho, Wo)
    ref = np.zeros_like(x)
    ref[:, ::s_, ::s_] = g
    assert np.array_equal(host(dx), ref)


@pytest.mark.parametrize("N,H,W,co,r", [(2, 5, 6, 3, 4), (2, 4, 13, 21, 8), (1, 3, 5, 7, 3), (1, 2, 3, 600, 8), (2, 3, 9, 5, 1)])
def test_phase_shift(L, N, H, W, co, r):
    """Subpixel._phase_shift (subpixel.py:77-88) and its inverse, bit-exact (a permutation): the LDS-transposing kernel
    (full and ragged pixel blocks, the SegModel head's 21 x 8 x 8, odd r) and the generic fallback (a pixel too large
    for the LDS tile: co = 600 at r = 8)"""
    rng = np.random.default_rng(12)
    x = rng.normal(0, 1, (N, H, W, co * r * r)).astype(np.float32)
    ref = O.phase_shift(x, r)
    y = empty(*ref.shape)
    call("dl3_phase_shift", ptr(dev(x)), ptr(y), N, H, W, co, r, 0)
    assert np.array_equal(host(y), ref)
    back = empty(*x.shape)
    call("dl3_phase_shift", ptr(y), ptr(back), N, H, W, co, r, 1)
    assert np.array_equal(host(back), x)


def _sample_weights(rng, labels, C, void_w):
    """temporal sample weights; void_w: NON-zero weights on the void pixels too (uniform weights handed to fit): the
    void rows must still give zero loss and zero gradient, and count in nnz (Keras' mean(w != 0))"""
    w = rng.uniform(0.5, 2, labels.shape)
    return (w if void_w else (labels < C) * w).astype(np.float32)


@pytest.mark.parametrize("C,void_w", [(21, False), (21, True), (40, False), (40, True)])
def test_softmax_argmax_xent(L, C, void_w):
    rng = np.random.default_rng(13)
    M = 5000
    x = rng.normal(0, 3, (M, C)).astype(np.float32)
    labels = rng.integers(0, C + 1, M).astype(np.float32)
    # rows whose true-class probability has left Keras' clip interval [1e-7, 1 - 1e-7] on either side: the loss is the
    # constant -log(bound) and tf.clip_by_value hands NO gradient to the logits (round 4)
    for i in range(20):
        labels[i] = i % C
        x[i, i % C] = x[i].max() + 40.0 if i >= 10 else x[i].min() - 40.0
    w = _sample_weights(rng, labels, C, void_w)
    p = empty(M, C)
    call("dl3_softmax_fwd", ptr(dev(x)), ptr(p), M, C)
    assert relerr(host(p), O.softmax(x.astype(np.float64))) < 1e-5
    am = torch.empty(M, dtype=torch.int32, device="cuda")
    call("dl3_argmax", ptr(dev(x)), am.data_ptr(), M, C)
    assert np.array_equal(host(am), x.argmax(-1))
    nnz = empty(1)
    call("dl3_count_nonzero", ptr(dev(w)), M, ptr(nnz))
    assert host(nnz)[0] == float((w != 0).sum())
    loss_ref, dl_ref, p_ref = O.loss_sparse_xent_ignoring_last_label(x.astype(np.float64)[None], labels[None],
                                                                     w.astype(np.float64)[None])
    P = L.dl3_rows_partials(M)
    dl, lp, probs = empty(M, C), empty(P), empty(M, C)
    call("dl3_softmax_xent", ptr(dev(x)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), ptr(probs), ptr(dl), ptr(lp), M, C)
    assert relerr(host(dl), dl_ref[0]) < 1e-5
    assert np.all(host(dl)[:20] == 0) and np.all(dl_ref[0][:20] == 0)
    assert abs(host(lp).astype(np.float64).sum() - loss_ref) < 1e-5 * abs(loss_ref)
    assert relerr(host(probs), p_ref[0]) < 1e-5

@pytest.mark.parametrize("N,H,W,C,r,void_w", [(2, 4, 13, 21, 8, False), (2, 4, 13, 21, 8, True), (1, 3, 5, 7, 3, False),
                                              (2, 5, 6, 3, 4, True), (1, 2, 9, 30, 2, False)])
def test_shuffle_softmax_xent(L, N, H, W, C, r, void_w):
    """dl3_shuffle_softmax_xent == the reference's phase shift (subpixel.py:77-88) followed by the loss (utils.py:127-130),
    with the gradient handed back in the layout of the Subpixel convolution's own output (full and ragged pixel blocks,
    the SegModel head's 21 x 8 x 8, odd r, C > 24)"""
    rng = np.random.default_rng(17)
    u = rng.normal(0, 2, (N, H, W, C * r * r)).astype(np.float32)
    M = N * H * r * W * r
    labels = rng.integers(0, C + 1, M).astype(np.float32)
    w = _sample_weights(rng, labels, C, void_w)
    shuffled = O.phase_shift(u.astype(np.float64), r).reshape(1, M, C)
    loss_ref, dl_ref, _ = O.loss_sparse_xent_ignoring_last_label(shuffled, labels[None], w.astype(np.float64)[None])
    # the inverse permutation of the reference gradient: du[n, ia, ib, ch*r*r + p*r + q] = dl[n, ia*r+q, ib*r+p, ch]
    dl4 = dl_ref.reshape(N, H, r, W, r, C)                       # [n, ia, q, ib, p, ch]
    du_ref = dl4.transpose(0, 1, 3, 5, 4, 2).reshape(N, H, W, C * r * r)   # [n, ia, ib, ch, p, q]
    nnz = empty(1)
    call("dl3_count_nonzero", ptr(dev(w)), M, ptr(nnz))
    P = L.dl3_shuffle_xent_partials(N, H, W, C, r)
    assert P > 0
    du, lp = empty(*u.shape), empty(P)
    call("dl3_shuffle_softmax_xent", ptr(dev(u)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), ptr(du), ptr(lp), N, H, W, C, r)
    assert relerr(host(du), du_ref) < 1e-5
    assert abs(host(lp).astype(np.float64).sum() - loss_ref) < 1e-5 * abs(loss_ref)
    # and the composition of the two existing entry points gives the same numbers
    y, dl = empty(N, H * r, W * r, C), empty(M, C)
    lp2 = empty(L.dl3_rows_partials(M))
    call("dl3_phase_shift", ptr(dev(u)), ptr(y), N, H, W, C, r, 0)
    call("dl3_softmax_xent", ptr(y), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), None, ptr(dl), ptr(lp2), M, C)
    back = empty(*u.shape)
    call("dl3_phase_shift", ptr(dl), ptr(back), N, H, W, C, r, 1)
    assert relerr(host(du), host(back)) < 1e-6
    assert L.dl3_shuffle_xent_partials(1, 2, 3, 600, 8) == 0   # a pixel too large for the LDS tile: caller falls back


@pytest.mark.parametrize("void_w", [False, True])
def test_fused_upsample_xent(L, void_w):
    """dl3_upsample_softmax_xent == resize_bilinear_fwd followed by softmax_xent"""
    rng = np.random.default_rng(16)
    N, Hi, Wi, Ho, Wo, C = 2, 8, 8, 64, 64, 21
    lo = rng.normal(0, 2, (N, Hi, Wi, C)).astype(np.float32)
    M = N * Ho * Wo
    labels = rng.integers(0, C + 1, M).astype(np.float32)
    w = _sample_weights(rng, labels, C, void_w)
    up = O.resize_bilinear_tf1(lo.astype(np.float64), Ho, Wo).reshape(1, M, C)
    loss_ref, dl_ref, p_ref = O.loss_sparse_xent_ignoring_last_label(up, labels[None], w.astype(np.float64)[None])
    nnz = empty(1)
    call("dl3_count_nonzero", ptr(dev(w)), M, ptr(nnz))
    P = L.dl3_rows_partials(M)
    dl, lp, probs = empty(M, C), empty(P), empty(M, C)
    call("dl3_upsample_softmax_xent", ptr(dev(lo)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), ptr(probs), ptr(dl), ptr(lp),
         N, Hi, Wi, Ho, Wo, C)
    assert relerr(host(dl), dl_ref[0]) < 1e-4
    assert relerr(host(probs), p_ref[0]) < 1e-4
    assert abs(host(lp).astype(np.float64).sum() - loss_ref) < 1e-5 * abs(loss_ref)


@pytest.mark.parametrize("void_w", [False, True])
@pytest.mark.parametrize("dims", [(2, 8, 8, 64, 64, 21), (1, 5, 7, 17, 23, 3), (3, 16, 16, 64, 64, 32), (2, 4, 4, 4, 4, 2),
                                  (2, 32, 32, 512, 512, 21),    # the benchmark's row (MobileNetV2: logits at 1/16): two pixels per thread
                                  (1, 64, 64, 256, 256, 21),    # Xception-style 1/4 logits: three float4 per thread and source row
                                  (2, 16, 32, 64, 300, 8),      # second pixel of some threads only
                                  (1, 8, 100, 16, 200, 32)])    # source rows too long for the prefetching form
def test_xent_fold_and_rows(L, dims, void_w, monkeypatch):
    """dl3_upsample_softmax_xent_fold + dl3_resize_bilinear_bwd_rows == the oracle's loss and the gradient it sends
    through the transposed legacy-bilinear resize, without the full-resolution dlogits"""
    N, Hi, Wi, Ho, Wo, C = dims
    rng = np.random.default_rng(26)
    lo = rng.normal(0, 2, (N, Hi, Wi, C))
    M = N * Ho * Wo
    labels = rng.integers(0, C + 1, M).astype(np.float32)
    w = _sample_weights(rng, labels, C, void_w)
    tape = O.Tape()
    up = O.resize_bilinear_tf1(lo, Ho, Wo, tape=tape)
    loss_ref, dl_ref, _ = O.loss_sparse_xent_ignoring_last_label(up.reshape(1, M, C), labels[None], w.astype(np.float64)[None])
    dlo_ref = tape.backward(up, dl_ref.reshape(up.shape))[id(lo)]
    nnz = empty(1)
    call("dl3_count_nonzero", ptr(dev(w)), M, ptr(nnz))
    P = L.dl3_xent_fold_partials(N, Ho)
    xfold, lp, dlo = empty(N, Ho, Wi, C), empty(P), empty(N, Hi, Wi, C)
    call("dl3_upsample_softmax_xent_fold", ptr(dev(lo)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), ptr(xfold), ptr(lp), N, Hi,
         Wi, Ho, Wo, C)
    call("dl3_resize_bilinear_bwd_rows", ptr(xfold), ptr(dlo), C, N, Hi, Wi, Ho, C, 0)
    assert abs(host(lp).astype(np.float64).sum() - loss_ref) < 1e-5 * abs(loss_ref)
    assert relerr(host(dlo), dlo_ref) < 1e-4
    # accumulate form, and agreement with the unfused pair of ops
    base = rng.normal(0, 1, (N, Hi, Wi, C)).astype(np.float32)
    acc = dev(base)
    call("dl3_resize_bilinear_bwd_rows", ptr(xfold), ptr(acc), C, N, Hi, Wi, Ho, C, 1)
    assert relerr(host(acc), base + dlo_ref) < 1e-4
    dl, lp2, dlo2 = empty(M, C), empty(L.dl3_rows_partials(M)), empty(N, Hi, Wi, C)
    call("dl3_upsample_softmax_xent", ptr(dev(lo)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), None, ptr(dl), ptr(lp2),
         N, Hi, Wi, Ho, Wo, C)
    call("dl3_resize_bilinear_bwd", ptr(dl), C, ptr(dlo2), C, N, Hi, Wi, Ho, Wo, C, 0, None, 0)
    assert relerr(host(dlo), host(dlo2)) < 1e-5
    # rows that request their operands one row ahead (the default where the row fits) == rows that request their own
    monkeypatch.setenv("DL3_XENT_PREF", "0")
    xf2, lp3 = empty(N, Ho, Wi, C), empty(P)
    call("dl3_upsample_softmax_xent_fold", ptr(dev(lo)), ptr(dev(labels)), ptr(dev(w)), ptr(nnz), ptr(xf2), ptr(lp3), N, Hi,
         Wi, Ho, Wo, C)
    assert np.array_equal(host(xf2), host(xfold)) and np.array_equal(host(lp3), host(lp))


def test_transpose_batched(L):
    rng = np.random.default_rng(31)
    shapes = [(16, 96), (960, 160), (33, 7), (1, 40), (288, 64)]
    src = [dev(rng.normal(0, 1, sh).astype(np.float32)) for sh in shapes]
    dst = [empty(sh[1], sh[0]) for sh in shapes]
    rows, t0 = [], 0
    for a, b, (r, c) in zip(src, dst, shapes):
        tx = (c + 31) // 32
        rows.append([a.data_ptr(), b.data_ptr(), r, c, t0, tx])
        t0 += tx * ((r + 31) // 32)
    desc = torch.tensor(rows, dtype=torch.int64, device="cuda")
    call("dl3_transpose_batched", desc.data_ptr(), len(rows), t0)
    for a, b in zip(src, dst):
        assert np.array_equal(host(b), host(a).T)
    one = empty(160, 960)
    call("dl3_transpose", ptr(src[1]), ptr(one), 960, 160)
    assert np.array_equal(host(one), host(dst[1]))


def test_adam_fill(L):
    rng = np.random.default_rng(14)
    n = 10001
    p, g = rng.normal(0, 1, n).astype(np.float32), rng.normal(0, 1, n).astype(np.float32)
    m, v = rng.normal(0, 0.1, n).astype(np.float32), rng.uniform(0, 0.1, n).astype(np.float32)
    pd, md, vd = dev(p), dev(m), dev(v)
    lr, b1, b2, eps, gs = 1e-3, 0.9, 0.999, 1e-8, 0.5
    call("dl3_adam_step", ptr(pd), ptr(dev(g)), ptr(md), ptr(vd), n, lr, b1, b2, eps, gs)
    gg = g * gs
    m2 = b1 * m + (1 - b1) * gg
    v2 = b2 * v + (1 - b2) * gg * gg
    assert relerr(host(md), m2) < 1e-6 and relerr(host(vd), v2) < 1e-6
    assert relerr(host(pd), p - lr * m2 / (np.sqrt(v2) + eps)) < 1e-6
    f = empty(1000)
    call("dl3_fill", ptr(f), 2.5, 1000)
    assert np.all(host(f) == 2.5)


def test_error_reporting(L):
    """bad arguments come back as a status + message, not a crash (include/dl3.h conventions)"""
    rc = L.dl3_pwconv_fwd(None, 4, None, None, 0, None, None, None, 4, 4, 4, 4, None, stream())
    assert rc == -1 and b"null" in L.dl3_last_error()
    x = empty(16, 6)
    rc = L.dl3_dwconv3x3_fwd(ptr(x), None, None, 0, ptr(x), ptr(x), 1, 4, 4, 6, 1, 1, 1, 1, 4, 4, None, 0, stream())
    assert rc == -4



@pytest.mark.parametrize("shape", [(2, 24, 20, 32, 64, 1), (1, 17, 19, 32, 64, 2), (2, 16, 16, 64, 32, 1),
                                   (1, 9, 11, 32, 32, 1), (3, 33, 40, 64, 64, 1), (1, 70, 66, 32, 64, 1)])
def test_conv3x3_mfma_route(L, shape):
    """dense 3x3 conv between 32/64-channel tensors on the matrix pipe without a column matrix (xception
    entry_flow_conv1_2, deeplabv3p.py:289): forward + BN partials, weight gradient, bwd-data + mask + add + BN-backward
    partials; strides 1 and 2, odd sizes, more than one 256-pixel step per workgroup"""
    N, H, W, Cin, Cout, stride = shape
    assert L.dl3_conv3x3_mfma_supported(Cin, Cout) == 1 and L.dl3_conv3x3_mfma_supported(3, 32) == 0
    rng = np.random.default_rng(16)
    Ho, pt, _ = O.same_pads(H, 3, stride, 1)
    Wo, pl, _ = O.same_pads(W, 3, stride, 1)
    x = rng.normal(0, 1, (N, H, W, Cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (3, 3, Cin, Cout)).astype(np.float32)
    s = rng.uniform(0.5, 1.5, Cin).astype(np.float32)
    t = rng.normal(0, 0.5, Cin).astype(np.float32)
    z = s * x.astype(np.float64) + t
    xin = np.maximum(z, 0)
    tape = O.Tape()
    wv = w.astype(np.float64)
    ref = O.conv2d(xin, wv, stride, pt, pl, Ho, Wo, tape=tape)
    P = L.dl3_conv3x3_mfma_partials(N, Ho, Wo)
    y, part = empty(N, Ho, Wo, Cout), empty(P, Cout, 2)
    xd, sd, td, wd = dev(x), dev(s), dev(t), dev(w)
    geom = (N, H, W, Cin, Cout, stride, pt, pl, Ho, Wo)
    call("dl3_conv3x3_mfma_fwd", ptr(xd), ptr(sd), ptr(td), 1, ptr(wd), ptr(y), *geom, ptr(part))
    assert relerr(host(y), ref) < TOL
    s1, s2 = fold_partials(part, P, Cout)
    assert relerr(s1, ref.sum((0, 1, 2))) < 1e-3 and relerr(s2, (ref ** 2).sum((0, 1, 2))) < 1e-3
    g = rng.normal(0, 1, ref.shape).astype(np.float32)
    cA, cB, cC = [rng.normal(0, 1, Cout).astype(np.float32) for _ in range(3)]
    yraw = host(y)
    dY = cA * g.astype(np.float64) + cB * yraw + cC
    grads = tape.backward(ref, dY)
    gd, cAd, cBd, cCd = dev(g), dev(cA), dev(cB), dev(cC)
    wsb = L.dl3_conv3x3_mfma_bwd_weight_workspace(N, H, W, Cin, Cout, stride, Ho, Wo)
    ws = empty((wsb + 3) // 4)
    dw = empty(3, 3, Cin, Cout)
    call("dl3_conv3x3_mfma_bwd_weight", ptr(xd), ptr(sd), ptr(td), 1, ptr(gd), ptr(y), ptr(cAd), ptr(cBd), ptr(cCd),
         ptr(dw), *geom, ptr(ws), wsb)
    assert relerr(host(dw), grads[id(wv)]) < 1e-3
    wT = empty(Cout, 9 * Cin)
    call("dl3_transpose", ptr(wd), ptr(wT), 9 * Cin, Cout)
    P2 = L.dl3_conv3x3_mfma_partials(N, H, W)
    dx, dpart = empty(N, H, W, Cin), empty(P2, Cin, 2)
    add = rng.normal(0, 1, (N, H, W, Cin)).astype(np.float32)
    mean = rng.normal(0, 1, Cin).astype(np.float32)
    invstd = rng.uniform(0.5, 2, Cin).astype(np.float32)
    call("dl3_conv3x3_mfma_bwd_data", ptr(gd), ptr(y), ptr(cAd), ptr(cBd), ptr(cCd), ptr(wT), ptr(dx), ptr(xd), ptr(sd),
         ptr(td), 1, ptr(dev(add)), ptr(dev(mean)), ptr(dev(invstd)), ptr(dpart), *geom)
    dx_ref = grads[id(xin)] * (z > 0) + add
    assert relerr(host(dx), dx_ref) < TOL
    d1, d2 = fold_partials(dpart, P2, Cin)
    assert relerr(d1, dx_ref.sum((0, 1, 2))) < 1e-3
    assert relerr(d2, (dx_ref * (x - mean) * invstd).sum((0, 1, 2))) < 1e-3
    # plain gradient operand (cA == NULL), no mask, no add, no statistics
    dx2 = empty(N, H, W, Cin)
    call("dl3_conv3x3_mfma_bwd_data", ptr(gd), None, None, None, None, ptr(wT), ptr(dx2), None, None, None, 0, None, None,
         None, None, *geom)
    tape2 = O.Tape()
    xv = x.astype(np.float64)
    ref2 = O.conv2d(xv, wv, stride, pt, pl, Ho, Wo, tape=tape2)
    assert relerr(host(dx2), tape2.backward(ref2, g.astype(np.float64))[id(xv)]) < TOL
    # same result as the direct vector-ALU kernel
    y2 = empty(N, Ho, Wo, Cout)
    call("dl3_conv3x3_fwd", ptr(xd), ptr(sd), ptr(td), 1, ptr(wd), ptr(y2), *geom, None)
    assert relerr(host(y2), host(y)) < TOL
    # too small a workspace is refused, unsupported channel counts too
    assert L.dl3_conv3x3_mfma_bwd_weight(ptr(xd), ptr(sd), ptr(td), 1, ptr(gd), ptr(y), ptr(cAd), ptr(cBd), ptr(cCd),
                                         ptr(dw), *geom, ptr(ws), 16, stream()) == -3
    assert L.dl3_conv3x3_mfma_fwd(ptr(xd), None, None, 0, ptr(wd), ptr(y), N, H, W, 24, Cout, stride, pt, pl, Ho, Wo, None,
                                  stream()) == -4


# ---------------------------------------------------------------------------------------
# either side of the network: dl3_prepare_targets (utils.py:375-402), dl3_seg_counts (utils.py:132-157) — bit-exact
# ---------------------------------------------------------------------------------------
def _label_batch(rng, B, HW, C, dtype):
    lab = rng.integers(0, C, (B, HW)).astype(np.int64)
    lab[rng.random((B, HW)) < 0.1] = 255                    # VOC void
    if B > 1:
        lab[1] = 255                                        # an image with no valid pixel
    if B > 2:
        lab[2] = 3                                          # a single class
        lab[2, : HW // 7] = C + 2                           # labels above C-1 are void too
    if B > 3:
        lab[3] = rng.choice([0, 5, C - 1], HW, p=[0.9, 0.09, 0.01])  # strongly unbalanced
    return lab.astype(dtype)


@pytest.mark.parametrize("B,HW,C,dtype", [(4, 1000, 21, np.uint8), (5, 64 * 64, 21, np.int32), (2, 7, 2, np.uint8),
                                          (3, 512 * 512, 21, np.uint8), (1, 300, 255, np.int32)])
def test_prepare_targets_bit_exact(L, B, HW, C, dtype):
    from dl3_amd import capi
    rng = np.random.default_rng(B * 1000 + HW)
    lab = _label_batch(rng, B, HW, C, dtype)
    if dtype == np.int32 and B > 4:
        lab[4, ::3] = -7                                    # negative labels (int32 only) count as void
    Yr, SWr, Hr = O.prepare_targets(lab, C)
    t = torch.from_numpy(lab).cuda()
    Y, SW = empty(B, HW), empty(B, HW)
    hist = torch.full((B, C + 1), -1, dtype=torch.int32, device="cuda")
    call("dl3_prepare_targets", t.data_ptr(), capi.LABEL_U8 if dtype == np.uint8 else capi.LABEL_I32, B, HW, C, ptr(Y),
         ptr(SW), hist.data_ptr())
    assert np.array_equal(host(hist), Hr)
    assert np.array_equal(host(Y), Yr[:, :, 0])
    assert np.array_equal(host(SW).view(np.uint32), SWr.view(np.uint32))    # same float32 bits
    # histogram only (Y == SW == NULL)
    hist.fill_(-1)
    call("dl3_prepare_targets", t.data_ptr(), capi.LABEL_U8 if dtype == np.uint8 else capi.LABEL_I32, B, HW, C, None,
         None, hist.data_ptr())
    assert np.array_equal(host(hist), Hr)


def test_prepare_targets_rejects_bad_arguments(L):
    from dl3_amd import capi
    t = torch.zeros(4, dtype=torch.uint8, device="cuda")
    h = torch.zeros(400, dtype=torch.int32, device="cuda")
    assert L.dl3_prepare_targets(t.data_ptr(), 0, 1, 4, 256, None, None, h.data_ptr(), stream()) == -1
    assert b"classes" in L.dl3_last_error()
    assert L.dl3_prepare_targets(t.data_ptr(), 7, 1, 4, 21, None, None, h.data_ptr(), stream()) == -1
    assert L.dl3_prepare_targets(None, 0, 1, 4, 21, None, None, h.data_ptr(), stream()) == -1
    with pytest.raises(capi.DL3Error):
        call("dl3_seg_counts", None, None, 1, 4, 21, h.data_ptr())


@pytest.mark.parametrize("B,HW,C", [(3, 1000, 21), (2, 512 * 512, 21), (4, 33, 2)])
def test_seg_counts_and_metrics_bit_exact(L, B, HW, C):
    from dl3_amd import utils as U
    rng = np.random.default_rng(HW)
    yt = rng.integers(0, C + 1, (B, HW)).astype(np.float32)                 # C = void
    yt[0][yt[0] == 1] = 0                                                   # class 1 absent from image 0
    probs = rng.random((B, HW, C)).astype(np.float32)
    pred = probs.argmax(-1).astype(np.int32)
    ref = O.seg_counts(pred, yt, C)
    counts = torch.full((B, 3, C), -1, dtype=torch.int32, device="cuda")
    pt = torch.from_numpy(pred).cuda()
    call("dl3_seg_counts", pt.data_ptr(), ptr(dev(yt)), B, HW, C, counts.data_ptr())
    got = host(counts)
    assert np.array_equal(got, ref)
    # the metric ratios computed from the device counts are the host metrics of the reference, to the last bit
    assert U.Jaccard_from_counts(got) == U.Jaccard(yt[:, :, None], probs)
    assert U.accuracy_from_counts(got) == U.sparse_accuracy_ignoring_last_label(yt[:, :, None], probs)
    assert U.Jaccard_from_counts(got) == O.jaccard(yt, probs)
