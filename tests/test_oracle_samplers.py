"""The oracle's Binomial and Gamma samplers against the closed-form
distributions (the only meaning 'matches the reference's native RNG' can have
where the reference cannot run): chi-square on the Binomial pmf, KS on the
Gamma cdf, at fixed seeds."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

from oracle import pyoracle

CNT = 120_000


def _binom(n, p, which=0, seed=123, binidx=3):
    L = pyoracle.lib()
    out = np.empty(CNT, np.int64)
    L.orc_binomial_many(seed, 0, CNT, 0, binidx, which, n, p, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out


@pytest.mark.parametrize("n,p", [(10, 0.3), (100, 0.05), (1000, 0.005), (50, 0.5), (2000, 0.5), (10000, 0.0524),
                                 (10000, 0.93), (1_000_000, 0.3), (20, 0.97), (3, 0.2), (100_000, 1e-5), (25, 0.4),
                                 (19, 0.5), (21, 0.5), (2_000_000_000, 0.25)])
def test_binomial_matches_pmf(n, p):
    out = _binom(n, p)
    assert out.min() >= 0 and out.max() <= n
    # bins of width ~sigma/8 (width 1 for narrow distributions)
    width = max(1, int(np.sqrt(n * p * (1 - p)) / 8))
    lo, hi = int(out.min()), int(out.max())
    edges = np.arange(lo, hi + width + 1, width)
    obs = np.histogram(out, bins=edges)[0].astype(float)
    exp = (stats.binom.cdf(edges[1:] - 1, n, p) - stats.binom.cdf(edges[:-1] - 1, n, p)) * CNT
    m = exp >= 10
    chi = ((obs[m] - exp[m]) ** 2 / exp[m]).sum()
    rest_e = CNT - exp[m].sum()
    if rest_e > 5:
        chi += (obs[~m].sum() - rest_e) ** 2 / rest_e
    dof = int(m.sum())
    assert stats.chi2.sf(chi, dof) > 1e-4, (n, p, chi, dof)
    assert abs(out.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / CNT) + 1e-12


def test_binomial_edges():
    L = pyoracle.lib()
    assert L.orc_binomial(1, 2, 3, 4, 0, 0, 0.5) == 0
    assert L.orc_binomial(1, 2, 3, 4, 0, -5, 0.5) == 0
    assert L.orc_binomial(1, 2, 3, 4, 0, 1000, 0.0) == 0
    assert L.orc_binomial(1, 2, 3, 4, 0, 1000, 1.0) == 1000
    # the two draws of one bin use different words of the same block
    a = _binom(5000, 0.4, which=0)
    b = _binom(5000, 0.4, which=1)
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.02
    # reproducible
    assert (a == _binom(5000, 0.4, which=0)).all()


@pytest.mark.parametrize("k,th", [(0.02, 0.05), (0.5, 1.0), (1.0, 2.0), (3.3, 0.3), (200.0, 0.0465), (46000.0, 0.09), (1e6, 0.1)])
def test_gamma_matches_cdf(k, th):
    L = pyoracle.lib()
    out = np.empty(CNT)
    L.orc_gamma_many(77, 0, CNT, 0, 5, k, th, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert (out >= 0).all()
    assert stats.kstest(out, "gamma", args=(k, 0, th)).pvalue > 1e-4
    assert abs(out.mean() - k * th) < 5 * th * np.sqrt(k / CNT)


def test_gamma_degenerate_shapes_are_zero():
    L = pyoracle.lib()
    # RVsum's k*N <= 0 branch (src/init.jl:73-77) and the Gamma(1e-100, 1e-100) fallback bins
    assert L.orc_gamma(1, 2, 3, 4, 0.0, 1.0) == 0.0
    assert L.orc_gamma(1, 2, 3, 4, 1e-100 * 1e6, 1e-100) == 0.0
    assert L.orc_gamma(1, 2, 3, 4, -1.0, 1.0) == 0.0
