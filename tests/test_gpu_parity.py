"""Parity of the HIP path (through the C ABI of libepi_gfx950.so) with the CPU
oracle on the same seeded inputs.  Integer results are compared bit for bit;
the only floating-point outputs (rewards, log-likelihood, ESS) to the stated
tolerances.  PARITY UNPINNED: the oracle is this build's restatement of the
reference (no reference outputs exist); see oracle/epi_oracle.c."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import random_states, reference_test_params
import epidemicmodeling_amd as em
from epidemicmodeling_amd import _lib
from epidemicmodeling_amd._lib import EpiError, as_f64p, as_i64p, check
from oracle import pyoracle

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def _belief(pomdp, st):
    b = em.ParticleCollection(em.interface.context_for(pomdp), len(st[0]))
    b.upload(*st)
    return b


def _eq(dev, ref):
    for name, x, y in zip("SIRT", dev, ref):
        assert x.shape == y.shape and (x == y).all(), f"{name} differs at {np.argwhere(x != y)[:5].tolist()}"


# ---------------------------------------------------------------------------
# samplers
# ---------------------------------------------------------------------------
def test_binomial_sampler_bit_exact():
    pomdp = em.make_pomdp(LOD=5)
    ctx = em.interface.context_for(pomdp)
    L = pyoracle.lib()
    rng = np.random.default_rng(0)
    cnt = 200_000
    n = np.concatenate([rng.integers(0, 50, cnt // 4), rng.integers(0, 5000, cnt // 4), rng.integers(0, 2_000_000, cnt // 4),
                        rng.integers(0, 2 ** 31 - 1, cnt // 4)]).astype(np.int64)
    p = rng.random(cnt)
    p[::7] = rng.random(len(p[::7])) * 1e-4
    p[::11] = 1 - rng.random(len(p[::11])) * 1e-3
    p[::101] = 0.0
    p[::103] = 1.0
    for which in (0, 1):
        out = np.empty(cnt, np.int64)
        check(ctx.lib.epi_sample_binomial(ctx.handle, cnt, as_i64p(n), as_f64p(p), 99, 7, 5, which, as_i64p(out)))
        ref = np.array([L.orc_binomial(99, j, 7, 5, which, int(n[j]), float(p[j])) for j in range(cnt)], dtype=np.int64)
        bad = np.flatnonzero(out != ref)
        assert bad.size == 0, (bad[:5], n[bad[:5]], p[bad[:5]], out[bad[:5]], ref[bad[:5]])


def test_gamma_sampler_bit_exact():
    pomdp = em.make_pomdp(LOD=5)
    ctx = em.interface.context_for(pomdp)
    L = pyoracle.lib()
    rng = np.random.default_rng(1)
    cnt = 200_000
    shape = np.concatenate([rng.random(cnt // 4), rng.random(cnt // 4) * 20, 10.0 ** rng.uniform(1, 7, cnt // 4),
                            10.0 ** rng.uniform(-12, 0, cnt // 4)])
    shape[::97] = 0.0
    scale = 10.0 ** rng.uniform(-3, 1, cnt)
    out = np.empty(cnt)
    check(ctx.lib.epi_sample_gamma(ctx.handle, cnt, as_f64p(shape), as_f64p(scale), 4242, 3, 9, as_f64p(out)))
    ref = np.array([L.orc_gamma(4242, j, 3, 9, float(shape[j]), float(scale[j])) for j in range(cnt)])
    bad = np.flatnonzero(out.view(np.uint64) != ref.view(np.uint64))
    assert bad.size == 0, (bad[:5], shape[bad[:5]], out[bad[:5]], ref[bad[:5]])


# ---------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("delay", [0, 1, 5])
def test_step_injected_bit_exact(delay):
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(10 + delay)
    n, H = 5000, pomdp.horizon
    st = random_states(pomdp, n, rng)
    a = rng.random(n)
    a[::10] = 0.0
    a[1::10] = 1.0
    S, I, R, T = st
    sympt = np.floor(rng.random((n, H)) * 0.3 * I).astype(np.int64)
    nt = np.floor((I - T.sum(axis=2)).astype(np.float64) * a[:, None]).astype(np.int64)
    pos = np.floor(rng.random((n, H)) * np.maximum(nt, 0)).astype(np.int64)
    gam = rng.gamma(2.0, 50.0, size=(n, H)) * (rng.random((n, H)) > 0.2)
    b = _belief(pomdp, st)
    ni = np.empty(n, np.int64); po = np.empty((n, H), np.int64); fl = C.c_uint32()
    check(b.ctx.lib.epi_step_injected(b.ctx.handle, b.handle, as_f64p(a), 1, as_i64p(sympt), as_i64p(pos), as_f64p(gam),
                                      as_i64p(ni), as_i64p(po), C.byref(fl)))
    exp, eni, epo, efl = orc.step_injected(st, a, sympt, pos, gam)
    _eq(b.download(), exp)
    assert (ni == eni).all() and (po == epo).all() and fl.value == efl


@pytest.mark.parametrize("pidx", range(6))
def test_step_batch_bit_exact(pidx):
    pomdp = reference_test_params()[pidx]
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(20 + pidx)
    n = 3000
    scale = [30, 3000, 20000][pidx % 3]
    st = random_states(pomdp, n, rng, scale=scale)
    a = rng.random(n)
    a_prev = rng.random(n)
    b = _belief(pomdp, st)
    o, r, fl = em.step_batch(b, a, a_prev, seed=SEED + pidx, step=3)
    exp, eo, er, eni, efl = orc.step_batch(st, a, a_prev, seed=SEED + pidx, step=3)
    _eq(b.download(), exp)
    assert (o == eo).all() and fl == efl
    assert (r.view(np.uint64) == er.view(np.uint64)).all()          # same IEEE operations: bitwise equal


@pytest.mark.parametrize("n", [1, 2048, 2049, 8176, 8177])
def test_step_batch_small_population_paths_bit_exact(n):
    """Populations on either side of the sizes up to which transfers go through the context's page-locked staging
    buffer (128 KB: 2048 particles of 16 rows for upload / download, 8176 particles for the step's outputs) and the
    step's outputs live in the context's device scratch: the same results as the oracle on both paths, also with a
    per-particle action (which takes device scratch too)."""
    pomdp = em.make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(n)
    st = random_states(pomdp, n, rng, scale=3000)
    b = _belief(pomdp, st)
    _eq(b.download(), st)
    for step, a in enumerate((0.3, rng.random(n))):
        a_prev = 0.1 if step == 0 else 0.3
        o, r, fl = em.step_batch(b, a, a_prev, seed=SEED, step=step)
        st, eo, er, eni, efl = orc.step_batch(st, np.broadcast_to(a, n).copy(), np.full(n, a_prev), seed=SEED, step=step)
        _eq(b.download(), st)
        assert (o == eo).all() and fl == efl
        assert (r.view(np.uint64) == er.view(np.uint64)).all()
    b.close()


def test_step_variates_reproduce_the_step():
    """The recorded variates, injected into the deterministic step, give the same result."""
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(31)
    n, H = 2000, pomdp.horizon
    st = random_states(pomdp, n, rng)
    b = _belief(pomdp, st)
    sy = np.empty((n, H), np.int64); po = np.empty((n, H), np.int64); ga = np.empty((n, H)); fl = C.c_uint32()
    check(b.ctx.lib.epi_step_variates(b.ctx.handle, b.handle, 0.3, SEED, 2, as_i64p(sy), as_i64p(po), as_f64p(ga), C.byref(fl)))
    dev = b.download()
    exp, _, _, _ = orc.step_injected(st, 0.3, sy, po, ga)
    _eq(dev, exp)
    exp2, _, _, _, _ = orc.step_batch(st, 0.3, seed=SEED, step=2)
    _eq(dev, exp2)


@pytest.mark.parametrize("delay", [0, 2])
def test_recorded_variates_drive_the_transliterated_reference_step(delay):
    """The check a Julia harness would make with the reference's own SimStep, with the statement-by-statement
    numpy transliteration of src/simulate.jl:99-151 standing in for it (no Julia here): the device takes a
    Philox-driven step and hands back the 3 H variates every particle drew; the transliterated step, fed those
    draws in the reference's order (sympt[1..H], pos[1..H], gamma[1..H]), must arrive at the same integer
    compartments, pending tests and new infections."""
    from oracle import step_numpy as sn
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    rng = np.random.default_rng(77 + delay)
    n, H, a = 400, pomdp.horizon, 0.27
    st = random_states(pomdp, n, rng, scale=5000)
    b = _belief(pomdp, st)
    sy = np.empty((n, H), np.int64); po = np.empty((n, H), np.int64); ga = np.empty((n, H)); fl = C.c_uint32()
    check(b.ctx.lib.epi_step_variates(b.ctx.handle, b.handle, a, SEED, 5, as_i64p(sy), as_i64p(po), as_f64p(ga), C.byref(fl)))
    S2, I2, R2, T2 = b.download()
    b.close()
    S, I, R, T = st
    for j in range(n):
        eS, eI, eR, eT, eni, neg = sn.sim_step(int(S[j]), I[j], int(R[j]), T[j].T, a, pomdp.N, sy[j], po[j], ga[j])
        assert (S2[j], R2[j]) == (eS, eR), j
        assert (I2[j] == eI).all() and (T2[j].T == eT).all(), j
        assert I2[j][0] == eni


# ---------------------------------------------------------------------------
# initial beliefs, mean, sample
# ---------------------------------------------------------------------------
def test_initial_beliefs_bit_exact():
    pomdp = em.make_pomdp(LOD=5, test_delay=2)
    orc = pyoracle.Oracle(pomdp.to_c())
    b = em.initialbelief(pomdp, 10_000, seed=5)
    _eq(b.download(), orc.init_normal(10_000, 10_000.0, 5_000.0, 5))
    b2 = em.random_states(pomdp, 10_000, seed=6)
    _eq(b2.download(), orc.init_simplex(10_000, 6))


def test_initial_belief_larger_than_the_population_is_refused():
    """State(Normal(mu, sigma), params) asserts sum(I0) <= N (src/typedef.jl:164): 14 bins of about 10^5 in a
    population of 10^6 violate it."""
    pomdp = em.make_pomdp(LOD=5)
    with pytest.raises(EpiError, match="typedef.jl:164") as e:
        em.initialbelief(pomdp, 1000, mu=100_000.0, sigma=1_000.0, seed=3)
    assert e.value.code == -1


def test_belief_mean_and_sample():
    pomdp = em.make_pomdp(LOD=5, test_delay=2)
    orc = pyoracle.Oracle(pomdp.to_c())
    rng = np.random.default_rng(41)
    st = random_states(pomdp, 7777, rng)
    b = _belief(pomdp, st)
    m = em.mean(b)
    eS, eI, eR, eT, bad = orc.mean(st)
    assert (m.S, m.R) == (eS, eR) and (m.I == eI).all() and (m.Tests == eT.T).all() and not bad
    s = em.rand(b, seed=1, ctr=2)
    hits = np.flatnonzero((st[0] == s.S) & (st[2] == s.R) & (st[1] == s.I).all(axis=1))
    assert hits.size >= 1


def test_many_samples_in_one_call_equal_single_calls():
    """rand(b, count): element j is the particle a single call returns for (seed, ctr0 + j) -- on an uploaded belief and
    on one that an update left with a pending ancestor indirection (test_delay 2: pending-test rows included) -- and
    the draws are uniform over the particles."""
    pomdp = em.make_pomdp(LOD=5, test_delay=2)
    orc = pyoracle.Oracle(pomdp.to_c())
    st = random_states(pomdp, 5000, np.random.default_rng(43))
    b = _belief(pomdp, st)
    for updated in (False, True):
        if updated:
            _, mu, _ = orc.log_weights(st, 0.3, 0)
            em.update(em.BootstrapFilter(pomdp, 5000, seed=SEED), b, em.Action(0.3), int(np.median(mu)))
        many = em.rand_many(b, 300, seed=77, ctr0=1000)
        for j in (0, 1, 17, 150, 299):
            one = em.rand(b, seed=77, ctr=1000 + j)
            assert (one.S, one.R) == (many[j].S, many[j].R) and (one.I == many[j].I).all() and (one.Tests == many[j].Tests).all()
    S, I, R, T = b.download()
    idx = []
    for s in em.rand_many(b, 20_000, seed=5):
        hit = np.flatnonzero((S == s.S) & (R == s.R) & (I == s.I).all(axis=1))
        assert hit.size >= 1
        idx.append(hit[0])
    # resampled beliefs hold duplicates: compare the hit frequencies of the distinct particles with their multiplicities
    assert len(set(idx)) > 0.5 * len(np.unique(np.column_stack([S, R, I]), axis=0))
    with pytest.raises(EpiError):
        em.rand_many(b, 0)


# ---------------------------------------------------------------------------
# the filter update
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 1000, 2048, 2049, 50_000])
def test_pf_update_bit_exact(n):
    pomdp = em.make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    a = 0.2
    _, mu, _ = orc.log_weights(st, a, 0)
    o = int(np.median(mu)) + 3
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED, step=4)
    em.update(upd, b, em.Action(a), o)
    exp, info = orc.pf_update(st, a, o, SEED, 4)
    _eq(b.download(), exp)
    assert upd.last_flags == info["flags"]
    assert abs(upd.last_ess - info["ess"]) <= 1e-9 * info["ess"]
    F = pyoracle.lib().orc_frac_bits(n)
    ll = info["lwmax"] + np.log(info["W"] / 2.0 ** F / n) - 0.5 * np.log(2 * np.pi)
    assert abs(upd.last_loglik - ll) <= 1e-9 * max(1.0, abs(ll))


@pytest.mark.parametrize("delay", [0, 1, 5])
def test_pf_closed_loop_bit_exact(delay):
    """Several successive updates (the resampling gather is folded into the next
    step on the device; the oracle materialises every posterior)."""
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, a = 6000, 0.3
    st = orc.init_normal(n, 8_000.0, 4_000.0, SEED + delay)
    true = orc.init_normal(1, 8_000.0, 4_000.0, 777)
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    for day in range(6):
        true, o, _, _, _ = orc.step_batch(true, a, seed=555, step=day)
        em.update(upd, b, em.Action(a), int(o[0]))
        st, info = orc.pf_update(st, a, int(o[0]), SEED, day)
        assert upd.last_flags == info["flags"]
        if day in (0, 3, 5):
            _eq(b.download(), st)
        m = em.mean(b)
        eS, eI, eR, eT, bad = orc.mean(st)
        assert (m.S, m.R) == (eS, eR) and (m.I == eI).all() and (m.Tests == eT.T).all()


@pytest.mark.parametrize("delay", [0, 1])
def test_closed_loop_days_bit_exact(delay):
    """Several days of the closed loop per device call (true step, update on its observation handed over in device
    memory, posterior sums and order statistics) against the oracle taken day by day: observations, rewards, the
    true state, diagnostics, every posterior particle, the column sums and the order statistics."""
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 5000
    st = orc.init_normal(n, 8_000.0, 4_000.0, SEED + delay)
    tr = orc.init_normal(1, 8_000.0, 4_000.0, 777)
    b = _belief(pomdp, st)
    true = _belief(pomdp, tr)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    qs = np.array([0.05, 0.5, 0.95, 1.0])
    acts = [0.3, 0.3, 0.1, 0.4, 0.4]
    day, a_prev = 0, 0.0
    for chunk in (acts[:2], acts[2:]):                        # two calls: the second starts from a pending ancestor index
        res = em.interface.closed_loop_days(true, b, upd, chunk, 555, day, q=qs, want_sums=True)
        for j, a in enumerate(chunk):
            tr, o, r, _, fl = orc.step_batch(tr, a, a_prev=a_prev, seed=555, step=day)
            st, info = orc.pf_update(st, a, int(o[0]), SEED, day)
            assert res["o"][j] == o[0] and res["r"][j] == r[0]
            assert tuple(res["true_sir"][j]) == (tr[0][0], tr[1][0].sum(), tr[2][0])
            assert res["flags"][j] == (info["flags"] | fl)
            assert abs(res["ess"][j] - info["ess"]) <= 1e-9 * info["ess"]
            ll = info["lwmax"] + np.log(info["W"] / 2.0 ** pyoracle.lib().orc_frac_bits(n) / n) - 0.5 * np.log(2 * np.pi)
            assert abs(res["loglik"][j] - ll) <= 1e-9 * max(1.0, abs(ll))
            H = pomdp.horizon
            assert res["sums"][j][0] == st[0].sum() and res["sums"][j][1] == st[2].sum()
            assert (res["sums"][j][2:2 + H] == st[1].sum(axis=0)).all()
            assert (res["sums"][j][2 + H:].reshape(H, delay + 1) == st[3].sum(axis=0)).all()
            tot = np.sort(st[1].sum(axis=1))
            ranks = np.clip(np.ceil(qs * n).astype(np.int64), 1, n)
            assert (res["quant"][j] == tot[ranks - 1]).all()
            a_prev = a
            day += 1
        _eq(b.download(), st)
        _eq(true.download(), tr)
    assert upd.step == len(acts)


@pytest.mark.parametrize("N,mu", [(1500, 40.0), (2047, 60.0), (2048, 60.0), (300_000_000, 9_000_000.0)])
def test_closed_loop_order_statistics_for_small_and_large_populations(N, mu):
    """The first pass of the selection rides on the kernel of the column sums with the window [0, 2^bits(N)): a
    population below 2^11 is settled by that pass alone, one of 3 10^8 needs two more."""
    pomdp = em.make_pomdp(LOD=5, N=N)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 6001
    st = orc.init_normal(n, mu, mu / 2, 41)
    tr = orc.init_normal(1, mu, mu / 2, 42)
    b, true = _belief(pomdp, st), _belief(pomdp, tr)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    qs = np.array([0.0, 0.25, 0.5, 1.0])
    res = em.interface.closed_loop_days(true, b, upd, [0.3, 0.2], 555, 0, q=qs, want_sums=True)
    a_prev = 0.0
    for j, a in enumerate([0.3, 0.2]):
        tr, o, r, _, fl = orc.step_batch(tr, a, a_prev=a_prev, seed=555, step=j)
        st, info = orc.pf_update(st, a, int(o[0]), SEED, j)
        tot = np.sort(st[1].sum(axis=1))
        ranks = np.clip(np.ceil(qs * n).astype(np.int64), 1, n)
        assert (res["quant"][j] == tot[ranks - 1]).all(), (j, res["quant"][j], tot[ranks - 1])
        assert res["sums"][j][0] == st[0].sum() and (res["sums"][j][2:2 + pomdp.horizon] == st[1].sum(axis=0)).all()
        a_prev = a
    _eq(b.download(), st)


def test_closed_loop_days_rejects_bad_arguments():
    pomdp = em.make_pomdp(LOD=5)
    ctx = em.interface.context_for(pomdp)
    lib = ctx.lib
    b = em.initialbelief(pomdp, 500, seed=1)
    true2 = em.initialbelief(pomdp, 2, seed=2)
    true1 = em.initialbelief(pomdp, 1, seed=2)
    upd = em.BootstrapFilter(pomdp, 500, seed=1)
    with pytest.raises(ValueError, match="one particle"):
        em.interface.closed_loop_days(true2, b, upd, [0.2], 1, 0)
    acts = np.array([0.2])
    q5 = np.full(5, 0.5)
    quant = np.zeros(5, dtype=np.int64)

    def call(truth, belief, ndays, nq, q, quant_out):
        return lib.epi_closed_loop_days(ctx.handle, truth.handle, belief.handle, ndays, as_f64p(acts), 0.0, 1, 0, 1, 0, nq, as_f64p(q),
                                        None, None, None, None, None, None, None, as_i64p(quant_out))
    assert call(true2, b, 1, 0, None, None) == -1 and b"population of one" in lib.epi_last_error()
    assert call(true1, b, 0, 0, None, None) == -1 and b"ndays" in lib.epi_last_error()
    assert call(true1, b, 1, 5, q5, quant) == -1 and b"quantiles" in lib.epi_last_error()
    assert call(b, b, 1, 0, None, None) == -1
    q_bad = np.array([1.5])
    assert call(true1, b, 1, 1, q_bad, quant) == -1 and b"q[0]" in lib.epi_last_error()
    # nothing was advanced by the refused calls
    assert call(true1, b, 1, 0, None, None) == 0
    for x in (b, true1, true2):
        x.close()


def test_pf_million_particles_bit_exact():
    """Three successive updates of 10^6 particles against the oracle: every posterior particle, ancestor
    concentration included (the second and third updates resample a belief that already holds duplicates)."""
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, a = 1_000_000, 0.2
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    b = em.initialbelief(pomdp, n, seed=SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    for day in range(3):
        _, mu, _ = orc.log_weights(st, a, 0)
        o = int(np.percentile(mu, 40 + 10 * day))
        em.update(upd, b, em.Action(a), o)
        st, info = orc.pf_update(st, a, o, SEED, day)
        assert upd.last_flags == info["flags"] == 0
        assert 1_000 < info["ess"] < n and abs(upd.last_ess - info["ess"]) <= 1e-9 * info["ess"]
    _eq(b.download(), st)
    b.close()


def test_pf_degenerate_weights():
    pomdp = em.make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 3000
    st = orc.init_normal(n, 10_000.0, 5_000.0, 3)
    # a = 0 and o = 5: every weight is zero -> particles kept, flag raised
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    em.update(upd, b, em.Action(0.0), 5)
    exp, info = orc.pf_update(st, 0.0, 5, SEED, 0)
    assert upd.last_flags & _lib.EPI_F_ZERO_W and info["flags"] & 8
    _eq(b.download(), exp)
    # a = 0 and o = 0: every weight is one -> uniform resampling
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    em.update(upd, b, em.Action(0.0), 0)
    exp, info = orc.pf_update(st, 0.0, 0, SEED, 0)
    _eq(b.download(), exp)
    assert abs(upd.last_ess - n) < 1e-6
    # an observation far in the tail: a single particle takes (almost) all the weight
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    em.update(upd, b, em.Action(0.5), 10 ** 6)
    exp, info = orc.pf_update(st, 0.5, 10 ** 6, SEED, 0)
    _eq(b.download(), exp)
    assert upd.last_ess < 50


def test_pf_sharded_phases_equal_single_update():
    """Two shards in one process, exchanging through device buffers: the phased API
    gives the same posterior as the one-call update for every slot."""
    import torch
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    ctx = em.interface.context_for(pomdp)
    lib = ctx.lib
    n, G = 9001, 2
    st = orc.init_normal(n, 10_000.0, 5_000.0, 71)
    # make the weights lopsided so that particles cross the shard boundary
    st[1][: n // 2] //= 3
    st[0][:] = pomdp.N - st[1].sum(axis=1) - st[2]
    starts = [(n * g) // G for g in range(G + 1)]
    shards = []
    for g in range(G):
        m = starts[g + 1] - starts[g]
        b = em.ParticleCollection(ctx, m, halo_cap=m)
        b.upload(*(x[starts[g]:starts[g + 1]] for x in st))
        check(lib.epi_belief_set_shard(b.handle, n, starts[g]))
        shards.append(b)
    a, seed = 0.25, 1234
    _, mu, _ = orc.log_weights(st, a, 0)
    o = int(np.quantile(mu, 0.9))
    rows = lib.epi_state_rows(pomdp.horizon, pomdp.test_delay)
    for step in range(3):
        lm = []
        for b in shards:
            v = C.c_double()
            check(lib.epi_pf_predict(ctx.handle, b.handle, a, o, seed, step, C.byref(v)))
            lm.append(v.value)
        gmax = max(lm)
        w = (C.c_uint64 * G)()
        for g, b in enumerate(shards):
            wl, sq = C.c_uint64(), C.c_double()
            check(lib.epi_pf_weigh(ctx.handle, b.handle, gmax, C.byref(wl), C.byref(sq), None))
            w[g] = wl.value
        lo = (C.c_int64 * (G + 1))()
        check(lib.epi_pf_plan(n, G, w, seed, step, lo))
        for g, b in enumerate(shards):           # parents in shard g
            for h, dst in enumerate(shards):     # slots owned by shard h
                s0, s1 = max(lo[g], starts[h]), min(lo[g + 1], starts[h + 1])
                if s1 <= s0:
                    continue
                if g == h:
                    check(lib.epi_pf_select_local(ctx.handle, b.handle, G, g, w, seed, step, s0, s1))
                else:
                    buf = torch.empty(rows * (s1 - s0), dtype=torch.int32, device="cuda")
                    check(lib.epi_pf_export(ctx.handle, b.handle, G, g, w, seed, step, s0, s1, C.c_void_p(buf.data_ptr())))
                    torch.cuda.synchronize()
                    check(lib.epi_pf_import(ctx.handle, dst.handle, s0, s1, C.c_void_p(buf.data_ptr())))
        for b in shards:
            fl = C.c_uint32()
            check(lib.epi_pf_finish(ctx.handle, b.handle, C.byref(fl)))
        st, info = orc.pf_update(st, a, o, seed, step)
        got = [b.download() for b in shards]
        _eq(tuple(np.concatenate([g_[k] for g_ in got]) for k in range(4)), st)


def test_pf_late_imports_and_step_in_parts_equal_single_update():
    """The phased update left with its exchange open: finish before the received blocks are placed, then the next
    step in parts around the late imports.  Same posterior as the one-call update; the library refuses everything
    that would read slots still waiting for their particles."""
    import torch
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    ctx = em.interface.context_for(pomdp)
    lib = ctx.lib
    n, G = 9001, 2
    st = orc.init_normal(n, 10_000.0, 5_000.0, 71)
    st[1][: n // 2] = st[1][: n // 2] * 7 // 8       # a tilt: part of the first shard's slots go to the second's particles
    st[0][:] = pomdp.N - st[1].sum(axis=1) - st[2]
    starts = [(n * g) // G for g in range(G + 1)]
    shards = []
    for g in range(G):
        m = starts[g + 1] - starts[g]
        b = em.ParticleCollection(ctx, m, halo_cap=m)
        b.upload(*(x[starts[g]:starts[g + 1]] for x in st))
        check(lib.epi_belief_set_shard(b.handle, n, starts[g]))
        shards.append(b)
    a, seed = 0.25, 1234
    _, mu, _ = orc.log_weights(st, a, 0)
    o = int(np.quantile(mu, 0.6))
    rows = lib.epi_state_rows(pomdp.horizon, pomdp.test_delay)
    late = [[] for _ in range(G)]              # per shard: (slot range, packed block) not placed yet
    local = [(0, 0)] * G                       # per shard: the columns its own particles filled
    dl = torch.empty(G, dtype=torch.float64, device="cuda")
    parts_run = 0
    for step in range(4):
        lm = []
        for g, b in enumerate(shards):
            m = starts[g + 1] - starts[g]
            dptr = C.c_void_p(dl[g:].data_ptr())
            if not late[g]:
                check(lib.epi_pf_predict_dev(ctx.handle, b.handle, a, o, seed, step, dptr))
            else:
                c0, c1 = local[g]
                first, last = -(-c0 // 256) * 256, (c1 if c1 == m else (c1 // 256) * 256)
                # nothing may read the waiting slots
                assert lib.epi_pf_predict_dev(ctx.handle, b.handle, a, o, seed, step, dptr) == -5
                assert lib.epi_belief_sums(ctx.handle, b.handle, as_i64p(np.zeros(64, dtype=np.int64))) == -5
                assert lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, 0, m, dptr) == -5
                if last <= first:              # (nearly) every slot comes from the other shard: place them, then one step
                    for (s0, s1), buf in late[g]:
                        check(lib.epi_pf_import_late(ctx.handle, b.handle, s0, s1, C.c_void_p(buf.data_ptr())))
                    late[g] = []
                    check(lib.epi_pf_predict_dev(ctx.handle, b.handle, a, o, seed, step, dptr))
                    continue
                # parts hold whole workgroups
                assert lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, first + 1, 256, dptr) == -1
                check(lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, first, last - first, dptr))
                assert lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, first, last - first, dptr) == -5
                assert lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a + 0.1, o, seed, step, 0, first, dptr) == -1
                for (s0, s1), buf in late[g]:
                    assert lib.epi_pf_import_late(ctx.handle, b.handle, s0 - 1, s1, C.c_void_p(buf.data_ptr())) != 0
                    check(lib.epi_pf_import_late(ctx.handle, b.handle, s0, s1, C.c_void_p(buf.data_ptr())))
                    assert lib.epi_pf_import_late(ctx.handle, b.handle, s0, s1, C.c_void_p(buf.data_ptr())) == -5
                late[g] = []
                if first > 0:
                    check(lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, 0, first, dptr))
                if last < m:
                    check(lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, last, m - last, dptr))
                parts_run += 1
        torch.cuda.synchronize()
        gmax = float(dl.max().item())
        w = (C.c_uint64 * G)()
        for g, b in enumerate(shards):
            wl, sq = C.c_uint64(), C.c_double()
            check(lib.epi_pf_weigh(ctx.handle, b.handle, gmax, C.byref(wl), C.byref(sq), None))
            w[g] = wl.value
        lo = (C.c_int64 * (G + 1))()
        check(lib.epi_pf_plan(n, G, w, seed, step, lo))
        local = [(0, 0)] * G
        for g, b in enumerate(shards):
            for h, dst in enumerate(shards):
                s0, s1 = max(lo[g], starts[h]), min(lo[g + 1], starts[h + 1])
                if s1 <= s0:
                    continue
                if g == h:
                    check(lib.epi_pf_select_local(ctx.handle, b.handle, G, g, w, seed, step, s0, s1))
                    local[g] = (s0 - starts[g], s1 - starts[g])
                else:
                    buf = torch.empty(rows * (s1 - s0), dtype=torch.int32, device="cuda")
                    check(lib.epi_pf_export(ctx.handle, b.handle, G, g, w, seed, step, s0, s1, C.c_void_p(buf.data_ptr())))
                    late[h].append(((s0, s1), buf))
        for b in shards:
            check(lib.epi_pf_finish(ctx.handle, b.handle, None))
        st, info = orc.pf_update(st, a, o, seed, step)
    assert parts_run >= 1, parts_run
    # the posterior is readable once the last blocks are in place
    for g, b in enumerate(shards):
        if late[g]:
            with pytest.raises(Exception):
                b.download()
        for (s0, s1), buf in late[g]:
            check(lib.epi_pf_import_late(ctx.handle, b.handle, s0, s1, C.c_void_p(buf.data_ptr())))
    got = [b.download() for b in shards]
    _eq(tuple(np.concatenate([g_[k] for g_ in got]) for k in range(4)), st)


@pytest.mark.parametrize("delay", [0, 2])
def test_step_in_parts_of_a_fresh_population_equals_the_whole_step(delay):
    """epi_pf_predict_part_dev on a population without an ancestor index (just uploaded): three parts in scrambled order,
    the last one ragged, then the rest of the update; same posterior as the one-call update."""
    import torch
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 256 * 5 + 77
    st = orc.init_normal(n, 10_000.0, 5_000.0, 5)
    b = _belief(pomdp, st)
    ctx, lib = b.ctx, b.ctx.lib
    a, seed = 0.3, 99
    _, mu, _ = orc.log_weights(st, a, 0)
    o = int(np.median(mu))
    d = torch.full((5,), -1.0, dtype=torch.float64, device="cuda")
    for step in range(2):                       # the second update reads through the ancestor index
        for col0, count in ((512, n - 512), (0, 256), (256, 256)):
            check(lib.epi_pf_predict_part_dev(ctx.handle, b.handle, a, o, seed, step, col0, count, C.c_void_p(d.data_ptr())))
        check(lib.epi_pf_weigh_dev(ctx.handle, b.handle, C.c_void_p(d.data_ptr()), C.c_void_p(d[1:].data_ptr())))
        torch.cuda.synchronize()
        w = (C.c_uint64 * 1)(int(d[1].item()))
        check(lib.epi_pf_select_local(ctx.handle, b.handle, 1, 0, w, seed, step, 0, n))
        check(lib.epi_pf_finish(ctx.handle, b.handle, None))
        st, info = orc.pf_update(st, a, o, seed, step)
        _eq(b.download(), st)
        assert float(d[0].item()) == info["lwmax"]


# ---------------------------------------------------------------------------
# rollouts
# ---------------------------------------------------------------------------
def test_rollout_bit_exact():
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, T = 800, 12
    st = orc.init_simplex(n, 9)
    rng = np.random.default_rng(9)
    for actions in (0.4, rng.random(T), rng.random(n), rng.random((n, T))):
        b = _belief(pomdp, st)
        hist, rew, fl = em.rollout_batch(b, T, actions, seed=77, step0=5)
        exp, ehist, erew, efl = orc.rollout(st, T, actions, seed=77, step0=5)
        _eq(b.download(), exp)
        assert (hist == ehist).all() and fl == efl
        assert (rew.view(np.uint64) == erew.view(np.uint64)).all()


@pytest.mark.parametrize("n_start", [1, 37])
def test_rollout_score_bit_exact(n_start):
    """Scenario scoring: sequences x rollouts from a belief, discounted reward summed on the device,
    against the oracle's rollouts of the same trajectories reduced in the same order."""
    pomdp = em.make_pomdp(LOD=5, test_delay=1, test_period=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n_seq, R, T, disc = 23, 5, 9, 0.95
    st = orc.init_normal(n_start, 10_000.0, 5_000.0, 11)
    rng = np.random.default_rng(3)
    actions = rng.integers(0, 11, size=(n_seq, T)) / 10.0
    b = _belief(pomdp, st)
    score, fl = em.rollout_score(b, actions, R, discount=disc, seed=99, step0=4)
    _eq(b.download(), st)                                           # the belief is left as it was
    n = n_seq * R
    idx = np.arange(n) % n_start
    start = tuple(x[idx] for x in st)
    _, _, rew, efl = orc.rollout(start, T, np.repeat(actions, R, axis=0), a_prev0=0.0, seed=99, step0=4)
    acc = np.zeros(n)
    scale = 1.0
    for t in range(T):
        acc = scale * rew[:, t] if t == 0 else acc + scale * rew[:, t]
        scale *= disc
    exp = np.zeros(n_seq)
    for r in range(R):
        exp = exp + acc.reshape(n_seq, R)[:, r]
    exp = exp / R
    assert fl == efl
    assert (score.view(np.uint64) == exp.view(np.uint64)).all()
    planner = em.RandomShootingPlanner(pomdp, n_seq=64, rollouts_per_seq=4, horizon=6, seed=5)
    a = planner.action(b)
    assert 0.0 <= a.testing_prop <= 1.0 and planner.last_scores.shape == (64,)
    assert a.testing_prop == planner.last_sequences[int(np.argmax(planner.last_scores)), 0]


def test_reference_bounds_property_on_device():
    """test/InfectionSimTest.jl:14-30 through Simulate(): 0 <= S, sum(I), R <= N for
    7 parameter sets x 6 actions, T = 100."""
    for pidx, pomdp in enumerate(reference_test_params()):
        rng = np.random.default_rng(pidx)
        for a in np.arange(0, 1.01, 0.2):
            h = em.Simulate(100, em.State_random(pomdp, rng), pomdp, em.Action(float(a)), seed=pidx)
            arr = h.Array()
            assert arr.shape == (3, 100) and (arr >= 0).all() and (arr <= pomdp.N).all()
            assert (arr.sum(axis=0) == pomdp.N).all()


def test_gen_and_ensemble():
    pomdp = em.make_pomdp(LOD=5, test_period=3)
    orc = pyoracle.Oracle(em.unity_test_period(pomdp).to_c())
    s = em.State_from([2000] * 14, pomdp)
    sp, o, r = em.gen(pomdp, s, em.Action(0.3), seed=8, step=2)
    st = (np.array([s.S]), s.I[None, :], np.array([s.R]), s.Tests.T[None])
    rs, prev = 0.0, 0.0
    for i in range(3):
        st, eo, er, _, _ = orc.step_batch(st, 0.3, prev, seed=8, step=6 + i)
        rs, prev = rs + er[0], 0.3
    assert sp.S == st[0][0] and (sp.I == st[1][0]).all() and sp.R == st[2][0] and o == eo[0] and r == rs
    ens = em.SimulateEnsemble(30, 500, pomdp, em.Action(0.2), seed=3)
    assert ens.shape == (3, 30, 500) and (ens.sum(axis=0) == pomdp.N).all() and (ens >= 0).all()


# ---------------------------------------------------------------------------
# full-size properties (no oracle at these sizes)
# ---------------------------------------------------------------------------
def test_large_belief_properties():
    pomdp = em.make_pomdp(LOD=5)
    n = 2_000_000
    b = em.initialbelief(pomdp, n, seed=SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    for day in range(3):
        em.update(upd, b, em.Action(0.2), 9000 - 200 * day)
        assert upd.last_flags == 0 and 1 < upd.last_ess <= n
    S, I, R, T = b.download()
    assert (S + I.sum(axis=1) + R == pomdp.N).all()          # population conserved through predict + resample
    assert (I >= 0).all() and (R >= 0).all() and (S >= 0).all()
    # same seeds -> same posterior (counter-based streams)
    b2 = em.initialbelief(pomdp, n, seed=SEED)
    upd2 = em.BootstrapFilter(pomdp, n, seed=SEED)
    for day in range(3):
        em.update(upd2, b2, em.Action(0.2), 9000 - 200 * day)
    S2, I2, R2, T2 = b2.download()
    assert (S == S2).all() and (I == I2).all() and (R == R2).all()
    m = em.mean(b)
    assert m.S == int(np.rint(S.sum() / n)) and (m.I == np.rint(I.sum(axis=0) / n)).all()


# ---------------------------------------------------------------------------
# kernel variants: every implementation choice must leave the results unchanged
# ---------------------------------------------------------------------------
def _three_updates(pomdp, n):
    b = em.initialbelief(pomdp, n, seed=SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    info = []
    for day in range(3):
        em.update(upd, b, em.Action(0.2), 9000 - 300 * day)
        info.append((upd.last_ess, upd.last_flags))
    out = b.download()
    b.close()
    return out, info


@pytest.mark.parametrize("delay", [0, 1])
def test_step_implementations_and_screens_agree(delay, testing_lib):
    """Cooperative (LDS retry queues) vs one-thread-per-particle kernel, binary32 screens on vs
    off: four builds of the same update, bit-identical posteriors at 10^6 particles."""
    lib = testing_lib
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    n = 1_000_000
    try:
        ref, ref_info = _three_updates(pomdp, n)
        for coop, screen in ((1, 0), (0, 1), (0, 0)):
            check(lib.epi_debug_step_impl(coop))
            check(lib.epi_debug_screen(screen))
            got, info = _three_updates(pomdp, n)
            _eq(got, ref)
            assert info == ref_info
    finally:
        check(lib.epi_debug_step_impl(1))
        check(lib.epi_debug_screen(1))


def test_table_free_binomial_attempt_agrees_with_the_defining_test(testing_lib):
    """What the first phase of the cooperative kernel settles without the log-factorial table (squeeze, series bounds
    on log f(k) - log f(m)) is what the defining acceptance test gives: 2.7e7 attempts per case, from the smallest
    n p the rejection sampler takes to populations of 10^9, small and large standard deviations; and the bounds
    leave few attempts open where the standard deviation is not tiny."""
    ctx = em.interface.context_for(em.make_pomdp(LOD=5))
    out = (C.c_uint64 * 5)()
    cases = [(20, 40, 0.5), (25, 200, 0.4), (100, 2000, 0.3), (2000, 20000, 0.0051), (2000, 20000, 0.035), (2000, 20000, 0.127),
             (1000, 3000, 0.245), (500, 4000, 0.4401), (5000, 20000, 0.0025), (100000, 900000, 0.0001), (100000, 900000, 0.5),
             (1000000, 15000000, 0.003), (1000000, 15000000, 0.49), (4000000, 12000000, 0.000003), (1 << 30, 1 << 30, 0.37),
             (1 << 30, 1 << 30, 1.1e-8)]
    for k, (lo, span, pp) in enumerate(cases):
        check(ctx.lib.epi_debug_btrs_quick(ctx.handle, 1000 + k, lo, span, pp, 1650, out))
        tot, acc, rej, opn, bad = (int(v) for v in out)
        assert tot > 1e7 and acc + rej + opn == tot, (lo, span, pp, list(out))
        assert bad == 0, (lo, span, pp, list(out))
        assert acc > 0.7 * tot
        sigma = np.sqrt((lo + span / 2) * pp * (1 - pp))
        if sigma >= 8 and lo < 1 << 24:                    # (beyond 2^24 the binary32 arithmetic of the bounds is what leaves attempts open)
            assert opn < 0.02 * tot, (lo, span, pp, sigma, opn / tot)


def test_short_binary32_sqrt_and_quotient_are_correctly_rounded(testing_lib):
    """The device forms of the binary32 square root and quotient used by the samplers against
    the compiler's general expansions (what the host oracle's sqrtf and "/" compute), on EVERY
    binary32 value of the operand ranges: sqrt over [2^-30, 2^32), 1/x over [1, 2^20), and the
    quotient of the binary32 logarithm over its reduced argument range."""
    ctx = em.interface.context_for(em.make_pomdp(LOD=5))
    bad = C.c_uint64(1)
    for mode, lo, hi in ((0, 2.0 ** -30, np.nextafter(np.float32(2.0 ** 32), np.float32(0))),
                         (1, 1.0, np.nextafter(np.float32(2.0 ** 20), np.float32(0))),
                         (2, 0.70710677, 1.4142137)):
        check(ctx.lib.epi_debug_f32_forms(ctx.handle, mode, float(lo), float(hi), C.byref(bad)))
        assert bad.value == 0, (mode, bad.value)
    assert ctx.lib.epi_debug_f32_forms(ctx.handle, 3, 1.0, 2.0, C.byref(bad)) == -1        # EPI_EINVAL


def test_wide_state_retry_queue_overflow_bit_exact(testing_lib):
    """H = 56 infection ages: 112 binomial draws per particle, so a workgroup's rejected first
    attempts exceed the retry queue and the in-thread completion path runs; still bit-exact."""
    lib = testing_lib
    pomdp = em.make_pomdp(LOD=5, horizon=56, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 3000
    st0 = orc.init_normal(n, 10_000.0, 5_000.0, 5)
    _, mu, _ = orc.log_weights(st0, 0.3, 0)
    o = int(np.median(mu))
    try:
        for impl in (2, 0, 1):               # one thread per particle with the cooperative rounds, without them, then the default (four threads per particle)
            check(lib.epi_debug_step_impl(impl))
            st = st0
            b = _belief(pomdp, st)
            upd = em.BootstrapFilter(pomdp, n, seed=SEED)
            for step in range(3):
                em.update(upd, b, em.Action(0.3), o)
                st, info = orc.pf_update(st, 0.3, o, SEED, step)
                assert upd.last_flags == info["flags"]
            _eq(b.download(), st)
    finally:
        check(lib.epi_debug_step_impl(1))


@pytest.mark.parametrize("H,D,n", [(14, 1, 800), (18, 0, 257), (19, 3, 1100), (21, 0, 1500), (22, 0, 1500), (25, 1, 129), (27, 7, 640), (28, 0, 4001), (28, 2, 700), (29, 1, 1500), (43, 0, 900), (56, 0, 3000), (56, 3, 333), (64, 7, 450)])
def test_wide_state_bit_exact(H, D, n):
    """States of 18 to 28 infection ages have every particle stepped by two threads, wider ones by four, each owning every
    second (fourth) pair of infection terms and the age bins that seed them.  Random parameter sets at
    the edges of the sampler regimes, population sizes that leave the last workgroup partial: plain steps with
    per-particle outputs, filter updates (the observation moments are summed in bin order by one of the two threads),
    and rollouts with per-particle actions and the history rows, all against the oracle."""
    rng = np.random.default_rng(500 + H + D)
    N = int(rng.choice([5_000, 1_000_000, 300_000_000]))
    p = rng.random(H)
    p[rng.random(H) < 0.15] = 0.0
    p[rng.random(H) < 0.1] = 1.0
    p[rng.random(H) < 0.2] *= 1e-4
    shapes = 10.0 ** rng.uniform(-5, 1, H)
    shapes[rng.random(H) < 0.15] = 0.0
    scales = 10.0 ** rng.uniform(-3, 0.3, H)
    pomdp = em.make_pomdp(horizon=H, test_delay=D, N=N, pos_test_probs=list(p), Infdistributions=list(zip(shapes, scales)),
                          symptomatic_isolation_prob=float(rng.random()), asymptomatic_prob=float(rng.random()))
    orc = pyoracle.Oracle(pomdp.to_c())
    st = random_states(pomdp, n, rng, scale=max(1, N // (4 * H + 10)))
    a = float(rng.random())
    b = _belief(pomdp, st)
    exp = st
    for step in range(2 if D else 1):                                 # (the second step finds the test pipeline filled)
        o, r, fl = em.step_batch(b, a, a_prev=a if step else 0.0, seed=5, step=step)
        exp, eo, er, _, efl = orc.step_batch(exp, a, a if step else 0.0, seed=5, step=step)
        _eq(b.download(), exp)
        assert (o == eo).all() and fl == efl and (r.view(np.uint64) == er.view(np.uint64)).all()
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    _, mu, _ = orc.log_weights(exp, a, 0)
    obs = int(np.median(mu))
    for step in range(2):
        em.update(upd, b, em.Action(a), obs)
        exp, info = orc.pf_update(exp, a, obs, SEED, step)
        assert upd.last_flags == info["flags"]
        want = info["lwmax"] + np.log(info["W"] / 2.0 ** pyoracle.lib().orc_frac_bits(n) / n) - 0.5 * np.log(2 * np.pi)
        assert abs(upd.last_loglik - want) <= 1e-9 * max(1.0, abs(want))
    _eq(b.download(), exp)
    T = 4
    acts = rng.random((n, T))
    hist, rew, fl = em.rollout_batch(b, T, acts, seed=77, step0=3)
    exp2, ehist, erew, efl = orc.rollout(exp, T, acts, a_prev0=a, seed=77, step0=3)      # (the belief remembers the last action taken on it)
    _eq(b.download(), exp2)
    assert (hist == ehist).all() and fl == efl and (rew.view(np.uint64) == erew.view(np.uint64)).all()
    b.close()


def test_sharded_filter_over_rccl_single_rank(tmp_path):
    """ShardedFilter with the device shard and the nccl (RCCL) backend, world size 1: the
    collectives run on device tensors and the posterior equals the oracle's single update."""
    import torch
    import torch.distributed as dist
    from epidemicmodeling_amd.distributed import ShardedFilter
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, a = 20_000, 0.25
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    _, mu, _ = orc.log_weights(st, a, 0)
    obs = [int(np.quantile(mu, q)) for q in (0.8, 0.5)]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        f = ShardedFilter(pomdp, n, seed=SEED, device=0)
        f.init_normal(10_000.0, 5_000.0)
        exp = st
        for step, o in enumerate(obs):
            info = f.update(a, o)
            exp, einfo = orc.pf_update(exp, a, o, SEED, step)
            assert info["flags"] == einfo["flags"] and abs(info["ess"] - einfo["ess"]) <= 1e-9 * einfo["ess"]
            assert np.isfinite(info["loglik"])
        _eq(f.download_local(), exp)
        mS, mI, mR, mT = f.mean()
        eS, eI, eR, eT, _ = orc.mean(exp)
        assert mS == eS and mR == eR and (mI == eI).all()
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------
# ensembles kept on the device and the ODE-model losses
# ---------------------------------------------------------------------------
def test_ensemble_history_and_ode_losses():
    """The device ensemble equals the oracle's rollouts day for day; every trajectory's ODE loss equals the numpy
    restatement bit for bit (same operation order), the mean to 1e-12 relative (the reduction order differs)."""
    from oracle import odeloss_numpy as ol
    pomdp = em.make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    n, T = 700, 18
    st = orc.init_simplex(n, 21)
    rng = np.random.default_rng(4)
    acts = rng.random(n)
    b = _belief(pomdp, st)
    ens = em.Ensemble(b, T, acts, seed=31, step0=1)
    _, ehist, _, efl = orc.rollout(st, T, acts, seed=31, step0=1)
    arr = ens.Array()
    assert arr.shape == (3, T, n) and ens.flags == efl
    assert (arr == np.transpose(ehist[:, :, :3], (2, 1, 0))).all()
    for kind, p in (("SIR", [0.12, 0.31]), ("SIR_CTRL", [0.1, 0.3, 0.2]), ("SEIR", [0.12, 0.5, 0.8]),
                    ("SEIR_CTRL", [0.1, 0.4, 0.7, 0.3, 0.5])):
        for sub in (1, 8):
            got, pt = ens.loss(kind, p, substeps=sub, per_trajectory=True)
            ref = ol.per_trajectory_loss(ehist[:, :, :3], pomdp.N, em.odefit.KINDS[kind], p, acts, sub)
            assert (pt.view(np.uint64) == ref.view(np.uint64)).all(), kind
            assert abs(got - ref.sum() / n) <= 1e-12 * abs(got)
    with pytest.raises(_lib.EpiError):
        ens.loss("SIR", [0.1, 0.2, 0.3])
    ens.close()


def test_ode_fit_on_the_device_ensemble():
    """FitRandEnsemble / FitRandControlledEnsemble (src/ODEFit.jl:166-273): Nelder-Mead over the device loss
    ends below the loss of the first guess, with positive rates, reproducibly."""
    pomdp = em.make_pomdp(LOD=5)
    res, p = em.FitRandEnsemble("SIR", 30, 2000, pomdp, em.Action(0.1), seed=5)
    assert len(p) == 2 and res.fun < 0.5 and p[0] > 0 and p[1] > 0
    b = em.random_states(pomdp, 2000, seed=5)
    ens = em.Ensemble(b, 30, 0.1, seed=5)
    assert res.fun <= ens.loss("SIR", [0.1, 0.1]) and abs(ens.loss("SIR", p) - res.fun) <= 1e-15
    ens.close()
    res2, p2 = em.FitRandEnsemble("SIR", 30, 2000, pomdp, em.Action(0.1), seed=5)
    assert p2 == p
    res3, p3 = em.FitRandControlledEnsemble("SEIR", 20, 1000, pomdp, seed=6)
    assert len(p3) == 5 and np.isfinite(res3.fun)


def test_suminf_quantiles_and_belief_summaries():
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    for n in (1, 7, 50_001):
        st = orc.init_normal(n, 10_000.0, 5_000.0, 3)
        b = _belief(pomdp, st)
        upd = em.BootstrapFilter(pomdp, n, seed=SEED)
        _, mu, _ = orc.log_weights(st, 0.3, 0)
        em.update(upd, b, em.Action(0.3), int(np.median(mu)))          # leaves a pending ancestor indirection
        S, I, R, T = b.download()
        tot = np.sort(I.sum(axis=1))
        qs = np.array([0.0, 0.05, 0.5, 0.95, 1.0, 0.3333])
        got = em.suminf_quantiles(b, qs)
        ranks = np.clip(np.ceil(qs * n).astype(np.int64), 1, n)
        assert (got == tot[ranks - 1]).all(), (n, got, tot[ranks - 1])
    true = em.State_from((10_000.0, 5_000.0), pomdp, np.random.default_rng(0))
    # (a seed whose initial belief has a mean with R >= 0: every initial particle has R = 0, so the rounded means of S and
    # the I_i leave N - (S + sum I) = -2 ... 2, and the reference's mean asserts R >= 0, src/updater.jl:18, as the library does)
    b = em.initialbelief(pomdp, 20_000, seed=SEED + 1)
    h = em.SimulateClosedLoop(5, true, b, pomdp, lambda bel: em.Action(0.2), seed=9, keep_beliefs="summary")
    assert len(h.beliefs) == 5
    for s in h.beliefs:
        assert s.suminf_q[0] <= s.suminf_q[1] <= s.suminf_q[2] and s.suminf_q[0] <= int(s.mean.I.sum()) <= s.suminf_q[2]


# ---------------------------------------------------------------------------
# run-to-run reproducibility (workgroups hand draws to each other through LDS queues: a missing barrier shows up as a
# handful of particles in a hundred thousand that differ from one launch to the next)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("delay,H", [(0, 14), (1, 14), (0, 21)])
def test_step_and_update_repeat_bit_for_bit(delay, H):
    pomdp = em.make_pomdp(LOD=5, test_delay=delay) if H == 14 else em.make_pomdp(LOD=5, test_delay=delay, horizon=H)
    n = 300_007
    first = None
    for rep in range(5):
        b = em.initialbelief(pomdp, n, seed=11)
        o, r, _ = em.step_batch(b, 0.2, seed=77, step=3)
        upd = em.BootstrapFilter(pomdp, n, seed=5, step=4)
        em.update(upd, b, em.Action(0.3), int(np.median(o)))
        got = (o.copy(), r.copy()) + tuple(b.download()) + (np.array([upd.last_loglik, upd.last_ess]),)
        b.close()
        if first is None:
            first = got
        else:
            for name, x, y in zip(("o", "r", "S", "I", "R", "T", "diag"), first, got):
                assert (x == y).all(), (name, rep, int((x != y).sum()))


# ---------------------------------------------------------------------------
# error behaviour of the C ABI
# ---------------------------------------------------------------------------
def test_error_codes_and_messages():
    lib = _lib.load()
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    ctx = em.interface.context_for(pomdp)
    n = 100
    b = em.initialbelief(pomdp, n, seed=1)

    def code(call):
        rc = call()
        assert rc != 0 and lib.epi_last_error().decode()
        return rc

    bad = pomdp.to_c()
    bad.N = 2 ** 31                                                                 # counts are int32 on the device
    assert code(lambda: lib.epi_params_set(ctx.handle, C.byref(bad))) == -4        # EPI_ERANGE
    bad = pomdp.to_c()
    bad.p[3] = 1.5
    assert code(lambda: lib.epi_params_set(ctx.handle, C.byref(bad))) == -1        # EPI_EINVAL
    h = C.c_void_p()
    assert code(lambda: lib.epi_belief_create(ctx.handle, 0, 0, C.byref(h))) == -1
    assert code(lambda: lib.epi_ctx_create(10_000, C.byref(h))) == -1
    # phases out of order
    w, sq = C.c_uint64(), C.c_double()
    assert code(lambda: lib.epi_pf_weigh(ctx.handle, b.handle, 0.0, C.byref(w), C.byref(sq), None)) == -5     # EPI_ESTATE
    assert code(lambda: lib.epi_pf_finish(ctx.handle, b.handle, None)) == -5
    v = C.c_double()
    check(lib.epi_pf_predict(ctx.handle, b.handle, 0.2, 100, 1, 0, C.byref(v)))
    assert code(lambda: lib.epi_pf_update(ctx.handle, b.handle, 0.2, 100, 1, 0, None, None, None)) == -5
    check(lib.epi_pf_weigh(ctx.handle, b.handle, v.value, C.byref(w), C.byref(sq), None))
    assert code(lambda: lib.epi_pf_select_local(ctx.handle, b.handle, 1, 0, C.byref(w), 1, 0, 0, n + 1)) == -1   # slots not owned
    check(lib.epi_pf_select_local(ctx.handle, b.handle, 1, 0, C.byref(w), 1, 0, 0, n))
    assert code(lambda: lib.epi_pf_select_local(ctx.handle, b.handle, 1, 0, C.byref(w), 1, 0, 0, n)) == -5       # once per update
    # every phased call wants the belief's own context: none, or another shard's, is refused the same way everywhere
    other = C.c_void_p()
    check(lib.epi_ctx_create(0, C.byref(other)))
    buf = C.create_string_buffer(64)
    for wrong in (None, other):
        assert code(lambda: lib.epi_pf_export(wrong, b.handle, 1, 0, C.byref(w), 1, 0, 0, 1, buf)) == -1
        assert code(lambda: lib.epi_pf_import(wrong, b.handle, 0, 1, buf)) == -1
        assert code(lambda: lib.epi_pf_finish(wrong, b.handle, None)) == -1
        assert code(lambda: lib.epi_pf_weigh(wrong, b.handle, 0.0, C.byref(w), C.byref(sq), None)) == -1
        assert code(lambda: lib.epi_pf_weigh_dev(wrong, b.handle, buf, buf)) == -1
        assert code(lambda: lib.epi_belief_sums(wrong, b.handle, C.cast(buf, C.POINTER(C.c_int64)))) == -1
    check(lib.epi_ctx_destroy(other))
    check(lib.epi_pf_finish(ctx.handle, b.handle, None))
    # states that violate the stored layout (Tests row D+1 / column 1 must be zero) are rejected on upload
    S, I, R, T = b.download()
    T2 = T.copy()
    T2[0, 0, 0] = 1
    with pytest.raises(_lib.EpiError):
        b.upload(S, I, R, T2)
    # a sharded belief refuses the one-call update
    check(lib.epi_belief_set_shard(b.handle, 2 * n, n))
    assert code(lambda: lib.epi_pf_update(ctx.handle, b.handle, 0.2, 100, 1, 1, None, None, None)) == -5
    b.close()


# ---------------------------------------------------------------------------
# the benchmark configurations at full size (properties; the oracle does not reach these sizes in seconds)
# ---------------------------------------------------------------------------
def test_config2_closed_loop_million_particles_50_days():
    """10^6-particle belief, 50-day closed loop with a fixed action: population conserved in every particle,
    weights never degenerate, the belief tracks the true infected total, and the run is reproducible."""
    pomdp = em.make_pomdp()
    n, T = 1_000_000, 50

    def run():
        true = em.State_from((10_000.0, 5_000.0), pomdp, np.random.default_rng(2))
        b = em.initialbelief(pomdp, n, seed=SEED)
        h = em.SimulateClosedLoop(T, true, b, pomdp, lambda bel: em.Action(0.2), seed=SEED, keep_beliefs="summary")
        return h, b

    h, b = run()
    S, I, R, _ = b.download()
    assert (S + I.sum(axis=1) + R == pomdp.N).all() and (I >= 0).all() and (S >= 0).all() and (R >= 0).all()
    est = np.array([int(s.mean.I.sum()) for s in h.beliefs])
    lo = np.array([s.suminf_q[0] for s in h.beliefs])
    hi = np.array([s.suminf_q[2] for s in h.beliefs])
    assert (lo <= hi).all()
    assert ((lo <= est) & (est <= hi)).all()
    # after the first week the belief mean follows the true infected total
    assert np.abs(est[7:] - h.inf[7:]).max() < 0.25 * h.inf[7:].max()
    h2, b2 = run()
    assert (h2.inf == h.inf).all() and (h2.pos_test == h.pos_test).all()
    S2, I2, R2, _ = b2.download()
    assert (S2 == S).all() and (I2 == I).all() and (R2 == R).all()


def test_coarse_weight_flag_when_a_large_population_collapses():
    """3 10^7 particles carry 28 fractional weight bits.  An ordinary observation leaves the truncation bound n / W far
    below 10^-3; one far in the tail puts nearly all weight on a handful of particles, the bound passes 10^-3 and the
    update says so (EPI_F_COARSE_W) while still producing a valid posterior."""
    pomdp = em.make_pomdp()
    n = 30_000_000
    b = em.initialbelief(pomdp, n, seed=SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    em.update(upd, b, em.Action(0.2), 12_000)
    assert upd.last_flags == 0 and upd.last_ess > 1e5
    em.update(upd, b, em.Action(0.2), 60_000)
    assert upd.last_flags == _lib.EPI_F_COARSE_W, upd.last_flags
    assert 1.0 <= upd.last_ess < 1e4
    sums = np.zeros(2 + 14 + 14, dtype=np.int64)
    check(b.ctx.lib.epi_belief_sums(b.ctx.handle, b.handle, as_i64p(sums)))
    assert sums[:16].sum() == n * pomdp.N and (sums >= 0).all()
    b.close()


def test_config4_scenario_scoring_full_size():
    """10^5 action sequences x 64 rollouts (6.4 10^6 trajectories), T = 20: finite, reproducible scores; testing
    everyone every day costs more than testing nobody at the default loss weights only through the test term."""
    pomdp = em.make_pomdp()
    b = em.initialbelief(pomdp, 10_000, seed=SEED)
    rng = np.random.default_rng(0)
    n_seq, T = 100_000, 20
    seqs = rng.integers(0, 11, size=(n_seq, T)) / 10.0
    seqs[0], seqs[1] = 0.0, 1.0
    s1, f1 = em.rollout_score(b, seqs, 64, seed=7)
    s2, f2 = em.rollout_score(b, seqs, 64, seed=7)
    assert np.isfinite(s1).all() and (s1 == s2).all() and f1 == f2 and (s1 < 0).all()
    s3, _ = em.rollout_score(b, seqs[:2], 64, seed=7)
    assert (s3 == s1[:2]).all()                                   # a sequence's score does not depend on the batch


def test_config5_wide_state_ensemble():
    """The ensemble configuration at its full size, 10^7 trajectories over 30 days, for the default horizon H = 14
    (4.8 GB of history on the device); the wide states H = 28, 56 with 10^6 trajectories (their histories have the same
    shape, and 10^7 of them add nothing but run time).  A random action per trajectory; every day of every trajectory
    conserves the population (checked through the ODE loss of a model that cannot move anyone, and for H = 14 on the
    downloaded history of the first 10^6 trajectories' worth of memory)."""
    for H in (14, 28, 56):
        pomdp = em.make_pomdp(LOD=5, horizon=H)
        n, T = (10_000_000 if H == 14 else 1_000_000), 30
        b = em.random_states(pomdp, n, seed=H)
        acts = np.random.default_rng(H).random(n)
        ens = em.Ensemble(b, T, acts, seed=H)
        b.close()
        # alpha = beta = 0 keeps the model at its initial condition: the loss is then the ensemble's own mean squared
        # drift from day 1, a finite number that is identical for SIR and SEIR with gamma = 0 ...
        l_sir = ens.loss("SIR", [0.0, 0.0], substeps=1)
        l_seir = ens.loss("SEIR", [0.0, 0.0, 0.0], substeps=1)
        assert np.isfinite(l_sir) and l_sir == l_seir and l_sir >= 0
        arr = ens.Array() if H == 14 else None                  # int64[3, T, n]: 7.2 GB of host memory at 10^7
        if arr is not None:
            # random start states include degenerate ones (the reference asserts on them, src/simulate.jl:117); the
            # population is conserved wherever no flag was raised, i.e. for states inside the model's domain
            tot = arr.sum(axis=0)
            assert arr.shape == (3, T, n) and (tot[0] == pomdp.N).all()
        ens.close()


def test_generated_parameter_tables_through_the_device_path():
    """The deterministic parameter pipeline feeds the device: tables regenerated from a seed for a limit of detection
    and a horizon the built-in tables do not cover (LOD 4, 18 infection ages), then one step and two filter updates of
    2000 particles, device against oracle bit for bit."""
    pomdp = em.CovidPOMDP_generated(seed=11, n=20_000, LOD=4, horizon=18, test_delay=1)
    assert pomdp.horizon == 18 and len(pomdp.pos_test_probs) == 18 and 0.0 < max(pomdp.pos_test_probs) <= 1.0
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 2000
    st = orc.init_normal(n, 6_000.0, 3_000.0, 17)
    b = em.initialbelief(pomdp, n, mu=6_000.0, sigma=3_000.0, seed=17)
    _eq(b.download(), st)
    o, r, fl = em.step_batch(b, 0.35, seed=17, step=0)
    st, eo, er, _, efl = orc.step_batch(st, 0.35, seed=17, step=0)
    _eq(b.download(), st)
    assert (o == eo).all() and (r == er).all() and fl == efl
    upd = em.BootstrapFilter(pomdp, n, seed=17, step=1)
    for day in range(2):
        _, mu, _ = orc.log_weights(st, 0.35, 0)
        obs = int(np.quantile(mu, 0.3 + 0.4 * day))
        em.update(upd, b, em.Action(0.35), obs)
        st, info = orc.pf_update(st, 0.35, obs, 17, 1 + day)
        assert upd.last_flags == info["flags"] and abs(upd.last_ess - info["ess"]) <= 1e-9 * info["ess"]
        _eq(b.download(), st)
    b.close()


# ---------------------------------------------------------------------------
# randomised parameter sets: shapes, delays and probabilities at the edges of every sampler regime
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(12))
def test_random_parameter_sets_bit_exact(seed):
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.choice([1, 2, 3, 7, 14, 15, 16, 21]))
    D = int(rng.integers(0, 8))
    N = int(rng.choice([50, 5_000, 1_000_000, 300_000_000]))
    p = rng.random(H)
    p[rng.random(H) < 0.2] = 0.0
    p[rng.random(H) < 0.15] = 1.0
    p[rng.random(H) < 0.2] *= 1e-4                                   # inversion regime
    shapes = 10.0 ** rng.uniform(-6, 1, H)
    shapes[rng.random(H) < 0.2] = 0.0
    scales = 10.0 ** rng.uniform(-3, 0.5, H)
    pomdp = em.make_pomdp(horizon=H, test_delay=D, N=N, pos_test_probs=list(p), Infdistributions=list(zip(shapes, scales)),
                          symptomatic_isolation_prob=float(rng.random()), asymptomatic_prob=float(rng.random()))
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 1500
    st = random_states(pomdp, n, rng, scale=max(1, N // (4 * H + 10)))
    a = float(rng.random())
    b = _belief(pomdp, st)
    # plain steps (per-particle outputs) ...
    o, r, fl = em.step_batch(b, a, seed=5, step=1)
    exp, eo, er, _, efl = orc.step_batch(st, a, 0.0, seed=5, step=1)
    _eq(b.download(), exp)
    assert (o == eo).all() and fl == efl and (r.view(np.uint64) == er.view(np.uint64)).all()
    # ... and filter updates on top
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    _, mu, _ = orc.log_weights(exp, a, 0)
    obs = int(np.median(mu))
    for step in range(2):
        em.update(upd, b, em.Action(a), obs)
        exp, info = orc.pf_update(exp, a, obs, SEED, step)
        assert upd.last_flags == info["flags"]
    _eq(b.download(), exp)


@pytest.mark.parametrize("mode", ["mixed", "tiny", "deep"])
def test_vanishing_test_probabilities_bit_exact(mode):
    """Test probabilities down to the subnormals: the variance terms n p (1 - p) then lie below the range in which a
    square root can do without rescaling its argument, and parameter sets like these take the kernel that rescales.
    `mixed`: ordinary bins dominate the variance; `tiny`: every term is below 2^-767; `deep`: every term is a
    subnormal number or close to one.  The largest log weight and the fixed-point weight total (which
    every log weight enters) are compared exactly, then two updates through the resampled belief."""
    H = 14
    p = np.array([0.4, 1e-250, 3e-300, 0.7, 5e-324, 2.2e-308, 1e-211, 0.05, 1e-230, 0.0, 1.0, 1e-305, 0.3, 1e-290])
    if mode == "tiny":
        p = np.where(p > 1e-200, 1e-240, p)
    elif mode == "deep":
        p = np.array([5e-324, 1e-320, 2.2e-308, 1e-310, 3e-312, 0.0, 3e-309, 4e-315, 1e-309, 6e-311, 2e-308, 1e-322, 7e-317, 9e-313])
    pomdp = em.make_pomdp(horizon=H, pos_test_probs=list(p))
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 3000
    st = random_states(pomdp, n, np.random.default_rng(77), scale=3000)
    a = 0.35
    _, mu, _ = orc.log_weights(st, a, 0)
    obs = int(np.median(mu))
    b = _belief(pomdp, st)
    lib, ctx = b.ctx.lib, b.ctx
    lwmax, wl, sq = C.c_double(), C.c_uint64(), C.c_double()
    check(lib.epi_pf_predict(ctx.handle, b.handle, a, obs, SEED, 0, C.byref(lwmax)))
    check(lib.epi_pf_weigh(ctx.handle, b.handle, lwmax.value, C.byref(wl), C.byref(sq), None))
    _, info = orc.pf_update(st, a, obs, SEED, 0)
    assert np.isfinite(lwmax.value) and wl.value > 0
    assert lwmax.value == info["lwmax"] and wl.value == info["W"]
    b.close()
    b = _belief(pomdp, st)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    exp = st
    for step in range(2):
        em.update(upd, b, em.Action(a), obs)
        exp, info = orc.pf_update(exp, a, obs, SEED, step)
        assert upd.last_flags == info["flags"]
        _eq(b.download(), exp)
    b.close()


def test_maximum_state_shape_bit_exact():
    """EPI_MAX_H = 64 infection ages with EPI_MAX_D = 7 days of test delay (507 state rows), population close
    to the int32 limit: plain steps, filter updates and the mean against the oracle."""
    H, D, N = 64, 7, 2_000_000_000
    rng = np.random.default_rng(77)
    p = np.clip(rng.random(H) * 1.2 - 0.1, 0.0, 1.0)
    shapes = 10.0 ** rng.uniform(-4, 0.5, H)
    pomdp = em.make_pomdp(horizon=H, test_delay=D, N=N, pos_test_probs=list(p), Infdistributions=list(zip(shapes, [0.05] * H)))
    assert em.interface.context_for(pomdp).lib.epi_state_rows(H, D) == 2 + H + D * (H - 1)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 700
    st = random_states(pomdp, n, rng, scale=N // (3 * H))
    b = _belief(pomdp, st)
    exp = st
    for step in range(3):                                             # fills the test pipeline rows
        o, r, fl = em.step_batch(b, 0.4, a_prev=0.4 if step else 0.0, seed=11, step=step)
        exp, eo, er, _, efl = orc.step_batch(exp, 0.4, 0.4 if step else 0.0, seed=11, step=step)
        assert (o == eo).all() and fl == efl and (r.view(np.uint64) == er.view(np.uint64)).all()
    _eq(b.download(), exp)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    _, mu, _ = orc.log_weights(exp, 0.4, 0)
    obs = int(np.median(mu))
    em.update(upd, b, em.Action(0.4), obs)
    exp, info = orc.pf_update(exp, 0.4, obs, SEED, 0)
    assert upd.last_flags == info["flags"]
    _eq(b.download(), exp)
    m = em.mean(b)
    eS, eI, eR, eT, bad = orc.mean(exp)
    assert not bad and m.S == eS and m.R == eR and (m.I == eI).all() and (m.Tests == eT.T).all()
    b.close()


# ---------------------------------------------------------------------------
# the device samplers against the closed-form distributions, at a sample size the CPU suite does not reach
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("n,p", [(17_000, 0.035), (3_400, 0.56), (2_300, 0.0014), (4_800, 0.0049), (1_000_000, 0.3), (40, 0.5)])
def test_device_binomial_distribution(n, p):
    """4 10^6 draws per case from the device sampler (the parameter regimes of the default model among them):
    chi-square against the Binomial pmf and the mean within five standard errors."""
    from scipy import stats
    pomdp = em.make_pomdp(LOD=5)
    ctx = em.interface.context_for(pomdp)
    cnt = 4_000_000
    nn = np.full(cnt, n, dtype=np.int64)
    pp = np.full(cnt, p)
    out = np.empty(cnt, np.int64)
    check(ctx.lib.epi_sample_binomial(ctx.handle, cnt, as_i64p(nn), as_f64p(pp), 777, 1, 2, 0, as_i64p(out)))
    assert out.min() >= 0 and out.max() <= n
    width = max(1, int(np.sqrt(n * p * (1 - p)) / 8))
    lo, hi = int(out.min()), int(out.max())
    edges = np.arange(lo, hi + width + 1, width)
    obs = np.histogram(out, bins=edges)[0].astype(float)
    exp = (stats.binom.cdf(edges[1:] - 1, n, p) - stats.binom.cdf(edges[:-1] - 1, n, p)) * cnt
    m = exp >= 10
    chi = ((obs[m] - exp[m]) ** 2 / exp[m]).sum()
    assert stats.chi2.sf(chi, int(m.sum())) > 1e-5, (n, p, chi, int(m.sum()))
    assert abs(out.mean() - n * p) < 5 * np.sqrt(n * p * (1 - p) / cnt)


@pytest.mark.parametrize("k,th", [(341.0, 0.0465), (40_000.0, 0.122), (1.6, 0.21), (0.35, 0.16), (3.0e6, 1.0)])
def test_device_gamma_distribution(k, th):
    from scipy import stats
    pomdp = em.make_pomdp(LOD=5)
    ctx = em.interface.context_for(pomdp)
    cnt = 2_000_000
    out = np.empty(cnt)
    check(ctx.lib.epi_sample_gamma(ctx.handle, cnt, as_f64p(np.full(cnt, k)), as_f64p(np.full(cnt, th)), 4242, 3, 9, as_f64p(out)))
    assert (out > 0).all()
    assert stats.kstest(out, "gamma", args=(k, 0, th)).pvalue > 1e-5
    assert abs(out.mean() - k * th) < 5 * th * np.sqrt(k / cnt)


# ---------------------------------------------------------------------------
# the whole step against an independent sampler stack (numpy's Binomial / Gamma generators driving the numpy
# transliteration of the Julia step): the kind of agreement that is possible with a native RNG
# ---------------------------------------------------------------------------
def test_trajectory_distributions_match_native_rng_simulation():
    from scipy import stats
    from oracle import step_numpy as sn
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    cp = pomdp.to_c()
    H, D = pomdp.horizon, pomdp.test_delay
    q, p = np.array(cp.q[:H]), np.array(cp.p[:H])
    k, th = np.array(cp.k[:H]), np.array(cp.theta[:H])
    a, T, n = 0.3, 8, 4000
    s0 = em.State_from([3000] * H, pomdp)
    # device: n trajectories from the same start state
    st = (np.full(n, s0.S), np.tile(s0.I, (n, 1)), np.full(n, s0.R), np.zeros((n, H, D + 1), dtype=np.int64))
    b = _belief(pomdp, st)
    hist, _, fl = em.rollout_batch(b, T, a, seed=2024, want_rew=False)
    dS, dI, dR, _ = b.download()
    dev = {"S": dS, "I": dI.sum(axis=1), "R": dR, "pos": hist[:, :, 3].sum(axis=1)}
    # numpy's own generators through the transliterated step
    rng = np.random.default_rng(7)
    ref = {key: np.empty(n, dtype=np.int64) for key in dev}
    for m in range(n):
        S, I, R, tests, tot = int(s0.S), s0.I.copy(), int(s0.R), np.zeros((D + 1, H), dtype=np.int64), 0
        for _ in range(T):
            sympt = rng.binomial(I, q)                                     # simulate.jl:58
            pos = rng.binomial(sn.num_tested(I, tests, a), p)              # simulate.jl:79
            I2, _, _, _ = sn.update_isolations(I, R, tests, sympt, pos)
            shape = k * np.roll(I2, 1)                                     # simulate.jl:145, init.jl:71-78
            gam = np.where(shape > 0, rng.gamma(np.maximum(shape, 1e-300), th), 0.0)
            S, I, R, tests, _, _ = sn.sim_step(S, I, R, tests, a, pomdp.N, sympt, pos, gam)
            tot += int(pos.sum())
        ref["S"][m], ref["I"][m], ref["R"][m], ref["pos"][m] = S, int(I.sum()), R, tot
    assert fl == 0
    for key in dev:
        x, y = dev[key].astype(float), ref[key].astype(float)
        assert stats.ks_2samp(x, y).pvalue > 1e-4, key
        se = np.sqrt(x.var() / n + y.var() / n)
        assert abs(x.mean() - y.mean()) < 5 * se, (key, x.mean(), y.mean(), se)


def test_posterior_moments_match_native_rng_filter():
    """Bootstrap filters with numpy's generators (predict through the transliterated step, Normal observation weight
    from the pre-step state, systematic resampling) against the device filter: the posterior mean of the infected
    total after four updates, over independent runs of both, has the same distribution."""
    from scipy import stats
    from oracle import step_numpy as sn
    pomdp = em.make_pomdp(LOD=5)
    cp = pomdp.to_c()
    H = pomdp.horizon
    q, p = np.array(cp.q[:H]), np.array(cp.p[:H])
    k, th = np.array(cp.k[:H]), np.array(cp.theta[:H])
    a, T, n, runs = 0.25, 4, 1200, 20
    obs = [5200, 5000, 4900, 4700]
    zeros = np.zeros((1, H), dtype=np.int64)

    def native_run(rng):
        x = rng.normal(10_000.0, 5_000.0, size=(n, H))
        while (x < 0).any():                                           # truncated, not clipped (src/typedef.jl:163)
            x[x < 0] = rng.normal(10_000.0, 5_000.0, size=int((x < 0).sum()))
        I = np.rint(x).astype(np.int64)
        S = pomdp.N - I.sum(axis=1)
        R = np.zeros(n, dtype=np.int64)
        for o in obs:
            nt = np.floor(I * a)
            mu = (nt * p).sum(axis=1)
            var = (np.sqrt(nt * p * (1 - p)) ** 2).sum(axis=1)
            lw = -0.5 * (o - mu) ** 2 / var - 0.5 * np.log(var)
            S2, I2, R2 = np.empty_like(S), np.empty_like(I), np.empty_like(R)
            for m in range(n):
                sympt = rng.binomial(I[m], q)
                pos = rng.binomial(nt[m].astype(np.int64), p)
                Iu, _, _, _ = sn.update_isolations(I[m], int(R[m]), zeros, sympt, pos)
                shape = k * np.roll(Iu, 1)
                gam = np.where(shape > 0, rng.gamma(np.maximum(shape, 1e-300), th), 0.0)
                S2[m], I2[m], R2[m], _, _, _ = sn.sim_step(int(S[m]), I[m], int(R[m]), zeros, a, pomdp.N, sympt, pos, gam)
            w = np.exp(lw - lw.max())
            c = np.cumsum(w)
            pos_u = (np.arange(n) + rng.random()) * (c[-1] / n)
            anc = np.minimum(np.searchsorted(c, pos_u, side="right"), n - 1)
            S, I, R = S2[anc], I2[anc], R2[anc]
        return I.sum(axis=1).mean()

    def device_run(seed):
        b = em.initialbelief(pomdp, n, seed=seed)
        upd = em.BootstrapFilter(pomdp, n, seed=seed)
        for o in obs:
            em.update(upd, b, em.Action(a), o)
        m = em.mean(b)
        b.close()
        return float(m.I.sum())

    nat = np.array([native_run(np.random.default_rng(100 + r)) for r in range(runs)])
    dev = np.array([device_run(9000 + r) for r in range(runs)])
    assert stats.ks_2samp(nat, dev).pvalue > 1e-3, (nat.mean(), dev.mean())
    assert stats.ttest_ind(nat, dev, equal_var=False).pvalue > 1e-3, (nat.mean(), dev.mean(), nat.std(), dev.std())


# ---------------------------------------------------------------------------
# a population whose state rows pass 2^31 elements
# ---------------------------------------------------------------------------
def test_population_beyond_two_to_the_31_state_elements(monkeypatch, capsys):
    """3·10^8 particles (42 GB of state): column sums after two steps of the whole population against the same slots
    stepped as six shards, then three filter updates with their invariants (tools/big_population.py; the same check
    passes at 1.5·10^9 particles, 210 GB, in 9 s)."""
    import runpy
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "big_population.py")
    monkeypatch.setattr(sys, "argv", [script, "3e8", "5e7"])
    runpy.run_path(script, run_name="__main__")
    assert "population conserved through three updates; ok" in capsys.readouterr().out


# ---------------------------------------------------------------------------
# the sharded configuration at its full size, all shards in this process
# ---------------------------------------------------------------------------
def test_config3_eight_shards_full_size_equal_the_single_update():
    """10^8 particles as 8 shards of 1.25 10^7 (the multi-GPU configuration, here with every shard on the one GPU),
    updated through the library's own sharded update (epi_mctx: no Python between the phases): after each of three
    updates the column sums equal those of the one-call update of an unsharded copy, so do log-likelihood, ESS and
    flags, and at the end every particle of the first and the last shard."""
    from epidemicmodeling_amd.distributed import NodeFilter
    pomdp = em.make_pomdp()
    ctx = em.interface.context_for(pomdp)
    lib = ctx.lib
    H, D = pomdp.horizon, pomdp.test_delay
    n, G, seed, a = 100_000_000, 8, SEED, 0.2
    nsums = 2 + H + (D + 1) * H
    starts = [(n * g) // G for g in range(G + 1)]
    single = em.ParticleCollection(ctx, n)
    check(lib.epi_belief_init_normal(single.handle, 10_000.0, 5_000.0, seed))
    node = NodeFilter(pomdp, n, [0] * G, seed=seed)
    node.init_normal()
    for step in range(3):
        o = 9000 - 200 * step
        ll, ess, fl = C.c_double(), C.c_double(), C.c_uint32()
        check(lib.epi_pf_update(ctx.handle, single.handle, a, o, seed, step, C.byref(ll), C.byref(ess), C.byref(fl)))
        got = node.update(a, o)
        assert fl.value == 0 and got["flags"] == 0
        assert got["loglik"] == ll.value and abs(got["ess"] - ess.value) <= 1e-9 * ess.value
        arr = (C.c_int64 * nsums)()
        check(lib.epi_belief_sums(ctx.handle, single.handle, arr))
        assert (node.sums() == np.array(arr, dtype=np.int64)).all()
    S, I, R, T = single.download()
    single.close()
    s, i, r, t = node.download()
    for g in (0, G - 1):
        sl = slice(starts[g], starts[g + 1])
        assert (s[sl] == S[sl]).all() and (i[sl] == I[sl]).all() and (r[sl] == R[sl]).all() and (t[sl] == T[sl]).all()
    node.close()


def test_entry_points_that_only_hosts_outside_the_package_call():
    """epi_pf_update_async + epi_pf_collect (bench.py's loop) against epi_pf_update; the event timing hooks; the shape
    and stream queries."""
    from epidemicmodeling_amd.odefit import Ensemble
    from epidemicmodeling_amd.distributed import NodeFilter
    pomdp = em.make_pomdp(LOD=5)
    n = 5000
    b1, b2 = em.initialbelief(pomdp, n, seed=SEED), em.initialbelief(pomdp, n, seed=SEED)
    ctx, lib = b1.ctx, b1.ctx.lib
    assert lib.epi_ctx_stream(ctx.handle) is not None                     # the context's own stream
    check(lib.epi_ctx_timing(ctx.handle, 1))
    for step, o in enumerate((9000, 8500, 9500)):
        check(lib.epi_pf_update_async(ctx.handle, b1.handle, 0.2, o, SEED, step))
    ll1, ess1, fl1 = C.c_double(), C.c_double(), C.c_uint32()
    check(lib.epi_pf_collect(ctx.handle, b1.handle, C.byref(ll1), C.byref(ess1), C.byref(fl1)))
    tot, cnt = C.c_double(), C.c_int64()
    check(lib.epi_ctx_timing_read(ctx.handle, C.byref(tot), C.byref(cnt)))
    check(lib.epi_ctx_timing(ctx.handle, 0))
    assert cnt.value == 3 and 0.0 < tot.value < 1000.0
    ll2, ess2, fl2 = C.c_double(), C.c_double(), C.c_uint32()
    for step, o in enumerate((9000, 8500, 9500)):
        check(lib.epi_pf_update(ctx.handle, b2.handle, 0.2, o, SEED, step, C.byref(ll2), C.byref(ess2), C.byref(fl2)))
    assert (ll1.value, ess1.value, fl1.value) == (ll2.value, ess2.value, fl2.value)
    _eq(b1.download(), b2.download())
    assert 0.0 < lib.epi_last_predict_ms(ctx.handle) < 1000.0             # the last (untimed) update's step kernel
    ens = Ensemble(b1, 7, 0.3, seed=5)
    en, eT = C.c_int64(), C.c_int()
    check(lib.epi_ensemble_shape(ens._h, C.byref(en), C.byref(eT)))
    assert (en.value, eT.value) == (n, 7)
    ens.close()
    f = NodeFilter(pomdp, 1234, [0, 0, 0], seed=SEED)
    assert lib.epi_mctx_shards(f._h) == 3 and lib.epi_mctx_size(f._h) == 1234
    f.close(); b1.close(); b2.close()
