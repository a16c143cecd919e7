"""The oracle's deterministic step against (1) the independent numpy
transliteration of the Julia source, (2) analytic known-answer cases, (3) the
only property the reference's own test asserts (test/InfectionSimTest.jl:18):
0 <= S, sum(I), R <= N over 100-day rollouts of 7 parameter sets x 6 actions."""
import numpy as np
import pytest

from conftest import random_states, reference_test_params
from epidemicmodeling_amd import make_pomdp
from oracle import pyoracle, step_numpy


def _oracle(pomdp):
    return pyoracle.Oracle(pomdp.to_c())


def _random_variates(pomdp, st, a, rng):
    """Plausible variates for injected steps: sympt <= I, pos <= tested, gamma >= 0."""
    S, I, R, T = st
    n, H = I.shape
    sympt = np.floor(rng.random((n, H)) * 0.3 * I).astype(np.int64)
    nt = np.floor((I - T.sum(axis=2)).astype(np.float64) * a).astype(np.int64)
    pos = np.floor(rng.random((n, H)) * np.maximum(nt, 0)).astype(np.int64)
    gam = rng.gamma(2.0, 50.0, size=(n, H)) * (rng.random((n, H)) > 0.2)
    return sympt, pos, gam


@pytest.mark.parametrize("delay", [0, 1, 2, 5])
@pytest.mark.parametrize("a", [0.0, 0.2, 0.37, 1.0])
def test_injected_step_equals_numpy_transliteration(delay, a):
    pomdp = make_pomdp(test_delay=delay, LOD=5)
    orc = _oracle(pomdp)
    rng = np.random.default_rng(100 + delay)
    st = random_states(pomdp, 300, rng)
    sympt, pos, gam = _random_variates(pomdp, st, a, rng)
    (S2, I2, R2, T2), ni, po, fl = orc.step_injected(st, a, sympt, pos, gam)
    S, I, R, T = st
    for j in range(len(S)):
        eS, eI, eR, eT, eni, neg = step_numpy.sim_step(int(S[j]), I[j], int(R[j]), T[j].T, a, pomdp.N, sympt[j], pos[j], gam[j])
        assert (S2[j], R2[j], ni[j]) == (eS, eR, eni)
        assert (I2[j] == eI).all()
        assert (T2[j].T == eT).all()
        assert (po[j] == pos[j]).all()
        assert S2[j] + I2[j].sum() + R2[j] == pomdp.N          # population is conserved
        if delay == 0:
            assert not T2[j].any()


def test_float_floor_cases():
    """The three Float64 -> Int floors (simulate.jl:78, :112, :44) on products that
    are not exact in binary64 (0.1*30 = 3.0000000000000004, (1-3/10)*10 = 7.000000000000001...)."""
    pomdp = make_pomdp(LOD=5, N=1000)
    orc = _oracle(pomdp)
    H = pomdp.horizon
    I = np.zeros((1, H), np.int64); I[0, 4] = 30; I[0, 5] = 10
    st = (np.array([1000 - 40]), I, np.array([0]), np.zeros((1, H, 1), np.int64))
    lw, mu, var = orc.log_weights(st, 0.1, 3)
    nt = step_numpy.num_tested(I[0], np.zeros((1, H), np.int64), 0.1)
    assert nt[4] == int(np.floor(30 * 0.1)) == 3 and nt[5] == 1
    assert mu[0] == nt[4] * pomdp.pos_test_probs[4] + nt[5] * pomdp.pos_test_probs[5]
    sympt = np.zeros((1, H), np.int64); sympt[0, 5] = 3
    pos = np.zeros((1, H), np.int64); pos[0, 5] = 10
    gam = np.zeros((1, H)); gam[0, 3] = 2.9999999999999996
    (S2, I2, R2, T2), ni, po, fl = orc.step_injected(st, 0.1, sympt, pos, gam)
    # floor((1 - 3/10) * 10) = floor(7.000000000000001) = 7 leave through the test pipeline
    assert int(np.floor((1 - 3 / 10) * 10)) == 7
    assert R2[0] == 3 + 7 and I2[0, 6] == 0
    assert ni[0] == int(np.floor((960 / (1000 - 10)) * 2.9999999999999996))


def test_pure_ageing_when_nothing_happens():
    """a = 0 and no symptomatic isolation, zero infectiousness: I shifts by one age, the oldest recover."""
    pomdp = make_pomdp(symptomatic_isolation_prob=0.0, LOD=5, Infdistributions=[(0.0, 1.0)] * 14)
    orc = _oracle(pomdp)
    rng = np.random.default_rng(5)
    st = random_states(pomdp, 50, rng)
    (S2, I2, R2, T2), o, r, ni, fl = orc.step_batch(st, 0.0, seed=9, step=1)
    S, I, R, T = st
    assert (S2 == S).all() and (I2[:, 1:] == I[:, :-1]).all() and (I2[:, 0] == 0).all()
    assert (R2 == R + I[:, -1]).all() and (o == 0).all() and fl == 0
    assert np.allclose(r, -(pomdp.inf_loss * I2.sum(axis=1) / pomdp.N))


def test_certain_tests_and_full_isolation():
    """p_i = 1, a = 1, D = 0: everybody not isolating by symptoms tests positive and is removed
    (scaled by 1 - sympt/I, floored, simulate.jl:112)."""
    pomdp = make_pomdp(pos_test_probs=[1.0] * 14, symptomatic_isolation_prob=1.0, asymptomatic_prob=0.0)
    orc = _oracle(pomdp)
    rng = np.random.default_rng(6)
    st = random_states(pomdp, 40, rng)
    (S2, I2, R2, T2), o, r, ni, fl = orc.step_batch(st, 1.0, seed=3, step=0)
    S, I, R, T = st
    assert (o == I.sum(axis=1)).all()
    assert fl == 0 and (I2 >= 0).all()
    assert (S2 + I2.sum(axis=1) + R2 == pomdp.N).all()


@pytest.mark.parametrize("pidx", range(7))
def test_reference_bounds_property(pidx):
    """test/InfectionSimTest.jl:14-30: all(0 <= Array(simHist) <= N), T = 100, 5 random
    State(param) x actions 0:0.2:1."""
    pomdp = reference_test_params()[pidx]
    orc = _oracle(pomdp)
    st = orc.init_simplex(5, seed=1000 + pidx)
    for ai, a in enumerate(np.arange(0, 1.01, 0.2)):
        out, hist, rew, fl = orc.rollout(st, 100, float(a), seed=17 + ai, step0=0)
        assert fl == 0
        assert (hist[:, :, :3] >= 0).all() and (hist[:, :, :3] <= pomdp.N).all()
        assert (hist[:, :, :3].sum(axis=2) == pomdp.N).all()
        assert (hist[:, :, 3] >= 0).all()
        assert np.isfinite(rew).all() and (rew <= 0).all()


def test_initial_states():
    pomdp = make_pomdp(LOD=5)
    orc = _oracle(pomdp)
    S, I, R, T = orc.init_normal(2000, 10_000.0, 5_000.0, seed=11)
    assert (I >= 0).all() and (R == 0).all() and (S + I.sum(axis=1) == pomdp.N).all() and not T.any()
    # truncated Normal(1e4, 5e3) on [0, inf): mean = mu + sigma * phi(-2)/(1 - Phi(-2))
    from scipy import stats
    tm = stats.truncnorm.mean(-2, np.inf, loc=1e4, scale=5e3)
    assert abs(I.mean() - tm) < 5 * 5e3 / np.sqrt(I.size)
    S, I, R, T = orc.init_simplex(4000, seed=12)
    assert (S >= 0).all() and (I >= 0).all() and (R >= 0).all()
    assert (S + I.sum(axis=1) + R == pomdp.N).all()
    # Dirichlet(1,1,1): each share has mean 1/3
    assert abs(S.mean() / pomdp.N - 1 / 3) < 0.02 and abs(R.mean() / pomdp.N - 1 / 3) < 0.02
