"""Pins the oracle's building blocks: Philox against the published known-answer vectors of Random123, the elementary
functions against libm, the ziggurat tables and the Marsaglia-Tsang constant and squeeze against the properties the
samplers' exactness rests on, and the series bounds the device uses in front of the binomial acceptance test against
the exact log-probabilities."""
import ctypes as C
import math

import numpy as np
from scipy import stats

from oracle import pyoracle


def _philox(c, k, rounds):
    L = pyoracle.lib()
    cc = (C.c_uint32 * 4)(*c); kk = (C.c_uint32 * 2)(*k); o = (C.c_uint32 * 4)()
    L.orc_philox4x32(cc, kk, rounds, o)
    return list(o)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds (the round function) ...
    assert _philox([0] * 4, [0] * 2, 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 10) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # ... and with 7 rounds, the count the sampler streams use
    assert _philox([0] * 4, [0] * 2, 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 7) == \
        [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]


def test_log_exp_accuracy():
    L = pyoracle.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.random(20000), rng.random(20000) * 1e6, 10.0 ** rng.uniform(-300, 300, 20000)])
    for x in xs:
        ref = math.log(x)
        assert abs(L.orc_log(float(x)) - ref) <= 4.5e-16 * max(abs(ref), 1e-3)
    for x in rng.uniform(-700, 5, 40000):
        assert abs(L.orc_exp(float(x)) / math.exp(x) - 1) < 5e-16
    assert L.orc_exp(-709.0) == 0.0 and L.orc_exp(float("-inf")) == 0.0 and L.orc_exp(0.0) == 1.0
    for x in rng.random(20000).astype(np.float32):
        if 0 < x < 0.99:
            assert abs(L.orc_logf(float(x)) - math.log(float(x))) < 2e-7 * abs(math.log(float(x)))


def test_box_muller_is_standard_normal():
    L = pyoracle.lib()
    rng = np.random.default_rng(1)
    ws = rng.integers(0, 2 ** 32, size=(60000, 2), dtype=np.uint64)
    z0, z1 = C.c_float(), C.c_float()
    zs = np.empty((len(ws), 2))
    for j, (a, b) in enumerate(ws):
        L.orc_box_muller(int(a), int(b), C.byref(z0), C.byref(z1))
        zs[j] = z0.value, z1.value
    # against the same transform in binary64
    a, b = ws[:, 0].astype(np.float64), ws[:, 1]
    r = np.sqrt(-2 * np.log(a * 2.0 ** -32 + 2.0 ** -33))
    ang = ((b >> 30).astype(np.float64) + 0.5) * np.pi / 2 + (((b & 0x3fffffff).astype(np.int64) - (1 << 29)) + 0.5) * 2.0 ** -30 * np.pi / 2
    big = r > 0.05                           # the binary32 radius is coarse only next to 0
    assert np.abs(zs[big, 0] - (r * np.cos(ang))[big]).max() < 2e-6
    assert np.abs(zs[big, 1] - (r * np.sin(ang))[big]).max() < 2e-6
    assert stats.kstest(zs[:, 0], "norm").pvalue > 1e-3
    assert stats.kstest(zs[:, 1], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(zs.T)[0, 1]) < 0.02


def test_expf_accuracy():
    L = pyoracle.lib()
    L.orc_expf.restype = C.c_float; L.orc_expf.argtypes = [C.c_float]
    rng = np.random.default_rng(2)
    for x in np.concatenate([rng.uniform(-87, 0, 30000), -10.0 ** rng.uniform(-8, 1.9, 10000)]).astype(np.float32):
        got = L.orc_expf(float(x))
        assert abs(got / math.exp(float(x)) - 1) < 3e-7, (x, got)
    assert L.orc_expf(0.0) == 1.0


def _zig_tables():
    L = pyoracle.lib()
    kn = np.zeros(256, np.uint32); wn = np.zeros(256, np.float32); fn = np.zeros(257, np.float64)
    L.orc_zig_table(kn.ctypes.data_as(C.c_void_p), wn.ctypes.data_as(C.c_void_p), fn.ctypes.data_as(C.c_void_p))
    return kn, wn, fn


def test_ziggurat_tables_have_equal_areas_and_nested_rectangles():
    """256 strips of equal area v under exp(-x^2/2): x_1 = r > x_2 > ... > x_256 = 0; strip k >= 1 is the rectangle
    [0, x_k] x [f(x_k), f(x_{k+1})], strip 0 the base rectangle plus the tail.  kn[k] is the largest 23-bit j whose
    candidate j wn[k] lies under the curve for the whole height of the strip (x < x_{k+1}), with a margin of 2^-22."""
    kn, wn, fn = _zig_tables()
    r, v = 3.6541528853610088, 4.92867323399e-3
    x = np.sqrt(-2.0 * np.log(fn[1:]))                        # x_1 .. x_256 from the tabulated f values
    assert abs(x[0] - r) < 1e-12 and x[-1] == 0.0 and (np.diff(x) < 0).all()
    area = x[:-1] * (fn[2:] - fn[1:-1])                       # strips 1 .. 255
    assert np.abs(area / v - 1).max() < 1e-9
    tail = math.sqrt(math.pi / 2) * math.erfc(r / math.sqrt(2))
    assert abs((r * fn[1] + tail) / v - 1) < 1e-9             # strip 0
    X = np.concatenate([[v / fn[1]], x[:-1]])                 # right edge of strip k's candidates
    assert np.abs(wn.astype(np.float64) / (X * 2.0 ** -23) - 1).max() < 1e-7
    inner = np.concatenate([[r], x[1:]])                      # strip 0's rectangle ends at r, strip k's at x_{k+1}
    assert (kn.astype(np.float64) * wn <= inner).all()
    assert ((kn.astype(np.float64) + 2 ** 23 * 2.0 ** -21) * wn >= inner * (1 - 1e-6)).all()
    assert kn.max() < 2 ** 23


def test_ziggurat_normals_are_standard_normal():
    L = pyoracle.lib()
    L.orc_zig_normal.restype = C.c_float
    L.orc_zig_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    kn, wn, fn = _zig_tables()
    rng = np.random.default_rng(3)
    ws = rng.integers(0, 2 ** 32, size=200_000, dtype=np.uint64)
    z = np.array([L.orc_zig_normal(11, j, 0, 3, 0, int(w)) for j, w in enumerate(ws)])
    fast = (ws >> 9) < kn[ws & 255]
    assert 0.98 < fast.mean() < 0.99                          # the share settled by one compare and one multiply
    zf = (ws[fast] >> 9).astype(np.float32) * wn[ws[fast] & 255]
    assert (np.abs(z[fast]) == zf).all() and ((z[fast] < 0) == ((ws[fast] & 256) != 0))[zf > 0].all()
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01 and abs((z ** 4).mean() - 3) < 0.1
    assert (np.abs(z) > 3.6541528).mean() == pytest_approx(2 * stats.norm.sf(3.6541528), 0.25)


def pytest_approx(value, rel):
    import pytest
    return pytest.approx(value, rel=rel)


def test_marsaglia_tsang_constant_stays_in_its_band():
    """c from two Newton steps in binary32: 1 <= 9 d c^2 <= 1 + 1.4e-5 for every shape the sampler can see (d = shape -
    1/3 with shape >= 1 after the boost, up to populations of 2^31 times the largest tabulated k)."""
    L = pyoracle.lib()
    L.orc_gamma_c.restype = C.c_float; L.orc_gamma_c.argtypes = [C.c_float]
    rng = np.random.default_rng(4)
    ds = np.concatenate([np.float32(2.0 / 3.0) + rng.random(20000).astype(np.float32) * 4, (10.0 ** rng.uniform(-0.17, 12, 60000)).astype(np.float32),
                         np.array([2.0 / 3.0, 1.0, 2.0, 4.0, 1e3, 1e6, 2.0 ** 31, 2.0 ** 40], np.float32)])
    for d in ds:
        c = float(L.orc_gamma_c(float(d)))
        e = 9.0 * float(d) * c * c - 1.0
        assert 0.0 <= e <= 1.4e-5, (d, c, e)


def test_gamma_squeeze_is_below_the_acceptance_function():
    """The fast acceptance u < squeeze(c, x) may only accept what the defining test log u < x^2/2 + d (1 - v + log v)
    accepts: squeeze <= exp(x^2/2 + d (1 - v + log v)) wherever t = 1 + c x > 0."""
    L = pyoracle.lib()
    L.orc_gamma_c.restype = C.c_float; L.orc_gamma_c.argtypes = [C.c_float]
    L.orc_gamma_squeeze.restype = C.c_float; L.orc_gamma_squeeze.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    worst = 1.0
    for _ in range(60000):
        shape = float(10.0 ** rng.uniform(0, 7))
        d = shape - 0.3333333333333333
        c = L.orc_gamma_c(float(np.float32(d)))
        x = float(np.float32(rng.normal() * (1 if rng.random() < 0.8 else 2.5)))
        t = 1.0 + float(c) * x
        if t <= 0:
            continue
        v = t ** 3
        acc = math.exp(min(0.0, 0.5 * x * x + d * (1.0 - v + math.log(v))))
        sq = float(L.orc_gamma_squeeze(c, x))
        assert sq <= acc + 1e-7 * acc, (shape, x, sq, acc)
        if abs(x) <= 1.0:
            worst = min(worst, sq / acc)
    assert worst > 0.9                                          # (and close to it in the bulk of the normal)


def test_series_bounds_of_the_binomial_acceptance_test():
    """What the device evaluates in front of the log-factorial table (btrs_bounds in epi_rng.hpp), restated in binary64:
    with x* = (n + 1) pp, m = floor(x*), F = x* - m, V = (n + 1) pp (1 - pp) and j = |k - m|,
        U0 - R - e2 <= log f(k) - log f(m) <= U0 + e2
    for U0 = -S1 / V + (Q - P) S2' / (2 V^2), S1 = j ((j + 1) / 2 - F'), S2' = j^2 (j / 3 + 1/2 - F'),
    R = (2 P^3 + Q^3) (j + 1)^4 / (12 V^3), e2 = 0.084 j / V^2, wherever j P / V < 1/2 ((P, Q, F') = (pp, 1 - pp, F) above
    the mode, (1 - pp, pp, 1 - F) below).  Checked against the exact log-probabilities (lgamma) over random (n, pp, k)."""
    from scipy.special import gammaln
    rng = np.random.default_rng(6)
    cnt = 400_000
    n = np.floor(10.0 ** rng.uniform(1.4, 7.3, cnt))
    pp = np.where(rng.random(cnt) < 0.5, rng.uniform(1e-6, 0.5, cnt), 10.0 ** rng.uniform(-6, -0.31, cnt))
    keep = n * pp >= 10
    n, pp = n[keep], pp[keep]
    qq = 1 - pp
    xs = (n + 1) * pp
    m = np.floor(xs); F = xs - m; V = xs * qq
    sig = np.sqrt(n * pp * qq)
    k = np.clip(np.rint(m + rng.normal(size=n.size) * 2.5 * sig), 0, n)
    d = k - m
    below = d < 0
    j = np.abs(d)
    P = np.where(below, qq, pp); Q = 1 - P; Fp = np.where(below, 1 - F, F)
    valid = (j * P / V < 0.5) & (j > 0)
    S1 = j * ((j + 1) / 2 - Fp)
    S2 = j * j * (j / 3 + 0.5 - Fp)
    U0 = -S1 / V + (Q - P) * S2 / (2 * V * V)
    R = (2 * P ** 3 + Q ** 3) * (j + 1) ** 4 / (12 * V ** 3)
    e2 = 0.084 * j / V ** 2
    Lx = gammaln(m + 1) + gammaln(n - m + 1) - gammaln(k + 1) - gammaln(n - k + 1) + d * np.log(pp / qq)
    tol = 1e-6 + 2e-15 * (gammaln(n + 1))                      # rounding of the four lgamma values
    lo_ok = (U0 - R - e2 - tol <= Lx) | ~valid
    hi_ok = (Lx <= U0 + e2 + tol) | ~valid
    assert valid.mean() > 0.9
    assert lo_ok.all(), (n[~lo_ok][:3], pp[~lo_ok][:3], k[~lo_ok][:3], (U0 - R - e2 - Lx)[~lo_ok][:3])
    assert hi_ok.all(), (n[~hi_ok][:3], pp[~hi_ok][:3], k[~hi_ok][:3], (Lx - U0 - e2)[~hi_ok][:3])
    # and the band is narrow where it matters: at two standard deviations, R of the order 1 / sigma^2
    near = valid & (j <= 2 * sig) & (sig > 20)
    assert np.median((R * sig ** 2)[near]) < 3.0
