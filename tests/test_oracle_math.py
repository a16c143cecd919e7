"""Pins the oracle's building blocks: Philox against the published
known-answer vectors of Random123, the elementary functions against libm."""
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
