"""Generates tests/golden/*.npz from the CPU oracle (oracle/dl3_oracle.py).

The reference cannot be imported here (TensorFlow/Keras absent) and ships no golden
vectors, so these fixtures pin the BUILD's oracle, not the reference: they catch regressions of the
oracle itself and give the GPU path a fixed target that does not need the oracle's full forward at
test time.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dl3_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def cfg1_case():
    """BASELINE.json configs[0]: Deeplabv3(backbone='mobilenetv2', input_shape=(128,128,3), classes=2, OS=16),
    single-image forward.  Weights: seed 1; image: seed 0; BN moving stats calibrated on that image."""
    kw = dict(backbone="mobilenetv2", input_shape=(128, 128, 3), classes=2, OS=16)
    params = O.init_params(O.param_shapes("mobilenetv2", 2), seed=1)
    x = np.random.default_rng(0).integers(0, 256, (1, 128, 128, 3)).astype(np.float32)
    # calibrate on a 4-image batch so that image_pooling_BN (1x1 map) sees a non-degenerate variance
    xc = np.random.default_rng(5).integers(0, 256, (4, 128, 128, 3)).astype(np.float32)
    params = O.calibrate_bn(params, xc, **kw)
    logits, _ = O.forward(params, x, **kw)
    return kw, params, x, logits


def main():
    kw, params, x, logits = cfg1_case()
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
    print("cfg1: logits range", flat.min(), flat.max(), "argmax ones", int(logits.argmax(-1).sum()))
    # per-op vectors (tiny)
    rng = np.random.default_rng(7)
    x = rng.normal(0, 1, (1, 6, 7, 4)).astype(np.float32)
    w = rng.normal(0, 1, (3, 3, 4)).astype(np.float32)
    out = {}
    for s, r in ((1, 1), (1, 2), (2, 1), (1, 5)):
        Ho, pt, _ = O.same_pads(6, 3, s, r)
        Wo, pl, _ = O.same_pads(7, 3, s, r)
        out["dw_s%d_r%d" % (s, r)] = O.depthwise3x3(x, w, s, r, pt, pl, Ho, Wo)
    out["resize_6x7_to_13x20"] = O.resize_bilinear_tf1(x, 13, 20)
    out["resize_6x7_to_48x56"] = O.resize_bilinear_tf1(x, 48, 56)
    I = rng.normal(0, 1, (1, 2, 3, 2 * 9)).astype(np.float32)
    out["phase_shift_r3"] = O.phase_shift(I, 3)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), x=x, w=w, I=I, **out)
    print("written")


if __name__ == "__main__":
    main()
