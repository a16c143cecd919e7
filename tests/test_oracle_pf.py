"""The oracle's particle-filter pieces: observation weights against the closed
form, fixed-point systematic resampling against the sequential two-pointer walk
of ParticleFilters.jl's LowVarianceResampler, belief mean against numpy, and
posterior sanity of a complete update."""
import ctypes as C

import numpy as np
from scipy import stats

from conftest import random_states
from epidemicmodeling_amd import make_pomdp, State, Action, observation
from oracle import pyoracle, step_numpy

_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def test_log_weights_match_normal_pdf():
    pomdp = make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(3)
    st = random_states(pomdp, 200, rng)
    a, o = 0.2, 9000
    lw, mu, var = orc.log_weights(st, a, o)
    S, I, R, T = st
    for j in range(len(S)):
        m, sd = observation(pomdp, State(int(S[j]), I[j], int(R[j]), T[j].T), Action(a))
        assert mu[j] == m and abs(np.sqrt(var[j]) - sd) <= 1e-12 * sd
        ref = stats.norm.logpdf(o, m, sd) + 0.5 * np.log(2 * np.pi)
        assert abs(lw[j] - ref) < 1e-9 * max(1.0, abs(ref))
    # a = 0: degenerate Normal(0, 0): weight 1 at o == 0, 0 elsewhere
    lw0, _, _ = orc.log_weights(st, 0.0, 0)
    assert (lw0 == 0.0).all()
    lw1, _, _ = orc.log_weights(st, 0.0, 1)
    assert np.isneginf(lw1).all()


def _resample(lw, seed=5, step=2):
    L = pyoracle.lib()
    n = lw.size
    F = L.orc_frac_bits(n)
    wq = np.empty(n, np.uint64)
    L.orc_quantize(n, lw.ctypes.data_as(_f64p), float(lw.max()), F, wq.ctypes.data_as(_u64p))
    Cc = np.empty(n, np.uint64)
    W = L.orc_cumsum(n, wq.ctypes.data_as(_u64p), 0, Cc.ctypes.data_as(_u64p))
    u = L.orc_resample_u(seed, step)
    a1 = np.empty(n, np.int64); a2 = np.empty(n, np.int64)
    L.orc_ancestors(n, Cc.ctypes.data_as(_u64p), W, u, 0, n, a1.ctypes.data_as(_i64p))
    L.orc_ancestors_walk(n, Cc.ctypes.data_as(_u64p), W, u, a2.ctypes.data_as(_i64p))
    return wq, Cc, W, u, a1, a2, F


def test_systematic_resampling_equals_sequential_walk():
    rng = np.random.default_rng(8)
    for n in [1, 2, 7, 1000, 4097]:
        for spread in [0.1, 3.0, 40.0]:
            lw = rng.normal(0, spread, n)
            if n > 3:
                lw[rng.integers(0, n, n // 4)] = -np.inf          # zero-weight particles are never chosen
            lw[rng.integers(0, n)] = max(lw.max(), 0.0)
            wq, Cc, W, u, a1, a2, F = _resample(lw)
            assert (a1 == a2).all()
            assert (np.diff(a1) >= 0).all() and a1.min() >= 0 and a1.max() < n
            assert (wq[a1] > 0).all()
            assert wq.max() == 2 ** F
            # offspring counts are within 1 of n w_i / W (the defining property of systematic resampling)
            cnt = np.bincount(a1, minlength=n)
            expect = wq.astype(np.float64) * n / float(W)
            assert (np.abs(cnt - expect) < 1.0 + 1e-6).all()
    assert 0.0 <= pyoracle.lib().orc_resample_u(1, 2) < 1.0


def _fp64_low_variance_walk(w, u):
    """ParticleFilters.jl's LowVarianceResampler on binary64 weights, without any quantisation: step = sum(w) / n,
    slot m takes the first particle whose running sum c_i (accumulated in index order, as `c += w[i]` does) reaches
    U_m = (m + u) step."""
    n = w.size
    c = np.cumsum(w)                                  # sequential accumulation in index order
    U = (np.arange(n) + u) * (c[-1] / n)
    return np.minimum(np.searchsorted(c, U, side="left"), n - 1)


def test_fixed_point_resampler_against_the_fp64_walk_when_the_weights_collapse():
    """10^6 particles whose weight sits on a few dozen of them (ESS < 100): the offspring counts of the fixed-point
    resampler against those of the sequential binary64 walk of the reference's resampler, for the library's number of
    fractional bits at this size (33) and for 19 bits, which puts 10^6 particles where 10^8 are at their 26 bits
    (n 2^-F = 1.9).  The truncation bound n / W is what the library reports as EPI_F_COARSE_W above 10^-3."""
    L = pyoracle.lib()
    rng = np.random.default_rng(12)
    n = 1_000_000
    lw = -0.5 * (rng.normal(0.0, 1.0, n) * 3e4) ** 2               # log weights: a needle, ESS of a few dozen
    lw[rng.integers(0, n)] = 0.0
    w = np.exp(lw - lw.max())
    ess = w.sum() ** 2 / (w * w).sum()
    assert ess < 100
    u = L.orc_resample_u(5, 2)
    want = np.bincount(_fp64_low_variance_walk(w, u), minlength=n)
    for F, coarse in ((L.orc_frac_bits(n), False), (19, True)):
        wq = np.empty(n, np.uint64)
        L.orc_quantize(n, lw.ctypes.data_as(_f64p), float(lw.max()), F, wq.ctypes.data_as(_u64p))
        Cc = np.empty(n, np.uint64)
        W = L.orc_cumsum(n, wq.ctypes.data_as(_u64p), 0, Cc.ctypes.data_as(_u64p))
        anc = np.empty(n, np.int64)
        L.orc_ancestors(n, Cc.ctypes.data_as(_u64p), W, u, 0, n, anc.ctypes.data_as(_i64p))
        got = np.bincount(anc, minlength=n)
        bound = n / float(W)                                       # relative weight the truncation can remove
        assert (bound > 1e-3) == coarse
        # weight actually lost to truncation, relative to the total
        lost = 1.0 - float(W) / (w.sum() * 2.0 ** F)
        assert 0.0 <= lost <= bound
        # offspring counts: every particle within one child of n w_i / sum(w) in both, so they differ by the lost share
        l1 = np.abs(got - want).sum() / n
        assert l1 <= 2.0 * lost + 4.0 * np.count_nonzero(want) / n, (F, l1, lost)
        heavy = np.argsort(w)[-10:]
        assert (np.abs(got[heavy] - want[heavy]) <= 1 + 2.0 * lost * want[heavy]).all()
        # first two moments of the offspring counts
        assert got.sum() == n == want.sum()
        assert abs((got.astype(float) ** 2).sum() - (want.astype(float) ** 2).sum()) <= (4.0 * lost + 1e-4) * (want.astype(float) ** 2).sum()
        if not coarse:
            assert l1 < 1e-3


def test_belief_mean_rounds_half_even():
    pomdp = make_pomdp(LOD=5, test_delay=2)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(4)
    S, I, R, T = random_states(pomdp, 1000, rng)
    aS, aI, aR, aT, bad = orc.mean((S, I, R, T))
    eS, eI, eR, eT = step_numpy.belief_mean(S, I, T, pomdp.N)
    assert (aS, aR) == (eS, eR) and (aI == eI).all() and (aT == eT).all() and not bad
    # ties: sums 1, 3, 5 over 2 particles -> 0, 2, 2
    I2 = np.zeros((2, 14), np.int64); I2[0, :3] = [1, 3, 5]
    S2 = np.array([pomdp.N - 9, pomdp.N]); R2 = np.zeros(2, np.int64)
    aS, aI, aR, aT, bad = orc.mean((S2, I2, R2, np.zeros((2, 14, 3), np.int64)))
    assert list(aI[:3]) == [0, 2, 2]


def test_full_update_posterior_moves_towards_observation():
    pomdp = make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, a = 4000, 0.2
    st = orc.init_normal(n, 10_000.0, 5_000.0, seed=21)
    lw, mu, var = orc.log_weights(st, a, 0)
    o = int(np.quantile(mu, 0.8))                 # an observation in the upper part of the prior predictive
    out, info = orc.pf_update(st, a, o, seed=33, step=0)
    assert info["flags"] == 0 and 1.0 < info["ess"] <= n
    anc = info["anc"]
    assert (np.diff(anc) >= 0).all()
    # resampled particles are the propagated ancestors
    prop, _, _, _, _ = orc.step_batch(st, a, seed=33, step=0)
    assert (out[0] == prop[0][anc]).all() and (out[1] == prop[1][anc]).all() and (out[2] == prop[2][anc]).all()
    # the posterior mean of the predicted observation mean moved towards o
    assert abs(mu[anc].mean() - o) < abs(mu.mean() - o)
    assert (out[0] + out[1].sum(axis=1) + out[2] == pomdp.N).all()
    # all weights zero: particles are kept
    out0, info0 = orc.pf_update(st, 0.0, 5, seed=33, step=0)
    assert info0["flags"] & 8 and (info0["anc"] == np.arange(n)).all()
