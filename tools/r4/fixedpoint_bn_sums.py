"""CPU study for DESIGN.md §7 ("fewer BatchNorm finalize launches"): could the per-workgroup partial sums of a BatchNorm
(sum y, sum y^2 per channel; today P fp32 partial rows folded by dl3_bn_finalize in double) be accumulated ORDER-INDEPENDENTLY
with 64-bit integer atomics instead, so that a consumer derives scale / shift from ONE row and no fold launch exists?
Integer addition is associative: the result is bit-reproducible whatever order the workgroups arrive in.  This script
measures what the fixed-point quantisation costs against today's fold on synthetic channels (numpy only).

  python tools/r4/fixedpoint_bn_sums.py
"""
import numpy as np

F1, F2 = 24, 16          # fractional bits of the sum / sum-of-squares accumulators (int64)
rng = np.random.default_rng(0)


def study(name, y, rows_per_wg):
    M = y.shape[0]
    P = M // rows_per_wg
    yw = y[:P * rows_per_wg].reshape(P, rows_per_wg).astype(np.float32)
    s1 = yw.sum(1, dtype=np.float32)                  # what a workgroup hands over today (fp32)
    s2 = (yw * yw).sum(1, dtype=np.float32)
    n = P * rows_per_wg
    ex1, ex2 = yw.astype(np.float64).sum(), (yw.astype(np.float64) ** 2).sum()
    fold1, fold2 = s1.astype(np.float64).sum(), s2.astype(np.float64).sum()       # dl3_bn_finalize today
    q1 = np.rint(s1.astype(np.float64) * 2.0 ** F1).astype(np.int64).sum()        # integer atomics, any order
    q2 = np.rint(s2.astype(np.float64) * 2.0 ** F2).astype(np.int64).sum()
    fx1, fx2 = q1 / 2.0 ** F1, q2 / 2.0 ** F2
    # two accumulators per statistic: whole part (F = 0) and the residue at 2^-40 — range 2^63, resolution 2^-40, both associative
    def two(s):
        s = s.astype(np.float64)
        c = np.rint(s)
        return int(c.astype(np.int64).sum()) + int(np.rint((s - c) * 2.0 ** 40).astype(np.int64).sum()) / 2.0 ** 40
    tw1, tw2 = two(s1), two(s2)

    def stats(a, b):
        m = a / n
        return m, max(b / n - m * m, 0.0)
    me, ve = stats(ex1, ex2)
    mf, vf = stats(fold1, fold2)
    mx, vx = stats(fx1, fx2)
    mt, vt = stats(tw1, tw2)
    head = int(max(abs(int(q1)), abs(int(q2)))).bit_length()
    rel = lambda v: abs(v - ve) / max(ve, 1e-30)
    print("%-30s P=%5d | mean err: fold %.1e one-acc %.1e two-acc %.1e | var rel err: fold %.1e one-acc %.1e two-acc %.1e | "
          "one-acc uses %2d of 63 bits" % (name, P, abs(mf - me), abs(mx - me), abs(mt - me), rel(vf), rel(vx), rel(vt), head))


M = 2 * 64 * 64
study("B=2 64x64, unit normal", rng.normal(0, 1, M), 32)
study("B=2 64x64, mean 30 std 0.1", rng.normal(30, 0.1, M), 32)
study("B=2 64x64, std 1e-3", rng.normal(0, 1e-3, M), 32)
M = 128 * 256 * 256
study("B=128 256x256, unit normal", rng.normal(0, 1, M), 16384)
study("B=128 256x256, mean 6 std 3", rng.normal(6, 3, M), 16384)
study("B=128 256x256, |y| ~ 1e3", rng.normal(0, 1e3, M), 16384)
