"""Parameter pipeline (host): regeneration of the reference's two missing data
files and the fits of src/init.jl.

Reference-held numbers used as fixtures:
  * pos_test_probs at LOD = 5 printed by notebooks/InfectionSimDemo.ipynb (cell 1);
  * the per-day 3-parameter Gamma fits printed by
    notebooks/DistTransformsForUnimodality.ipynb (cell 11) -- their means
    shape*scale + loc are the column means of the sample-mean table; the
    printed run used samples of 10 (its comment says so), so the variance of
    an individual is 10 * shape * scale^2;
  * alpha ~ 0.19535 printed by notebooks/InfectiousnessDistributions.ipynb.
"""
import numpy as np
import pytest
import scipy.stats as st

from epidemicmodeling_amd import paramfit as pf, params as P

NOTEBOOK_FITS = {          # python day -> (shape, loc, scale); python day d is infection age d + 1
    3: (5.301941002463371, -0.026693373199305207, 0.02349043333105298),
    4: (252.14962895325738, -0.9927822710133167, 0.005224801428991194),
    5: (36356.503738806336, -13.29150360799273, 0.0003772379850182588),
    6: (751.2113273429273, -1.2621874440457281, 0.0022353256207562158),
    7: (141.74118579326722, -0.30177974082314335, 0.0047059912522378336),
    8: (66.13846332671301, -0.1359205591893147, 0.006521673584381098),
    9: (40.94541091124748, -0.09946126391135396, 0.007832068440718964),
    10: (22.256873185308265, -0.05986567569607234, 0.00956520822283739),
    11: (10.71109772155187, -0.02704161456436714, 0.01159354951553948),
    12: (4.631369058533229, -0.008594554787355307, 0.013878530801002223),
    13: (1.5869510310185997, -0.0007652723360705862, 0.01839558128026778),
}


@pytest.fixture(scope="module")
def tables():
    return pf.generate_tables(seed=0, LOD=5, return_data=True)


def _trapezoid_reference(t1, t2, alpha, knots):
    """Scalar restatement of integrate_inf (distUtils.py:37-76): trapezoid rule
    over the curve's knots inside (t1, t2)."""
    t0, tpeak, tf, vpeak = knots
    t0n = t0 + 3 * (tpeak - t0) / (vpeak - 3)
    h = alpha * max(vpeak - 6, 0)

    def val(t):
        if t0n < t < tpeak:
            return (h - 0) / (tpeak - t0n) * (t - t0n)
        if tpeak < t < tf:
            return h + (0 - h) / (tf - tpeak) * (t - tpeak)
        return 0.0
    ts, vs = [t1], [val(t1)]
    for bt, bv in ((t0n, 0.0), (tpeak, h), (tf, 0.0)):
        if t1 < bt < t2:
            ts.append(bt)
            vs.append(bv)
    ts.append(t2)
    vs.append(val(t2))
    return sum((vs[i + 1] + vs[i]) * (ts[i + 1] - ts[i]) / 2 for i in range(len(ts) - 1))


def test_closed_form_integral_equals_trapezoid_over_knots():
    rng = np.random.default_rng(3)
    knots = pf.viral_load_knots(200, rng)
    got = pf.secondary_infections(knots, 0.2, 0, 20)
    for j in range(200):
        want = [_trapezoid_reference(i, i + 1, 0.2, knots[j]) for i in range(21)]
        np.testing.assert_allclose(got[j], want, rtol=1e-11, atol=1e-14)
    # the whole curve: area of the triangle (distUtils.py:27-29)
    a, b, c, h = pf._triangle(knots, 0.2)
    full = pf.secondary_infections(knots, 0.2, 0, 60).sum(axis=1)
    np.testing.assert_allclose(full, 0.5 * (c - a) * h, rtol=1e-11)


def test_viral_load_curve_follows_get_beta():
    knots = np.array([[3.0, 5.0, 12.0, 9.0]])
    v = pf.viral_loads(knots, 0, 20)[0]
    assert (v[:4] == 0).all()                        # t <= t0
    assert v[4] == pytest.approx(3 + 6 / 2 * 1)      # rising 3 -> 9 over (3, 5)
    assert v[5] == pytest.approx(9.0)                # the peak
    assert v[12] == pytest.approx(3.0)               # back at 10^3 at tf
    assert v[20] == 0.0 and (np.diff(v[5:]) <= 0).all()     # the same line continues, floored at 0


def test_scale_matches_notebook_alpha(tables):
    assert tables["alpha"] == pytest.approx(0.19535, rel=1e-2)
    # mean number of secondary infections over the whole course is R0
    k, th = np.array(tables["Infdistributions"]).T
    assert 2.40 < float((k * th).sum()) < 2.5       # days 15.. carry the remaining ~1 %


def test_pos_test_probs_match_notebook_table(tables):
    got = np.array(tables["pos_test_probs"])
    want = np.array(P.POS_TEST_PROBS[5])             # printed by the reference
    assert (got[:3] == 0).all()
    assert np.abs(got - want).max() < 5e-3           # two Monte-Carlo tables of 10^5 rows: s.e. of a difference <= 2.3e-3


def test_sample_mean_columns_match_notebook_moments(tables):
    m = tables["infections"]
    for d, (shape, loc, scale) in NOTEBOOK_FITS.items():
        assert m[:, d].mean() == pytest.approx(shape * scale + loc, rel=2e-2), d
    for d in range(4, 13):                           # variance of one individual (fits away from the ends)
        shape, loc, scale = NOTEBOOK_FITS[d]
        assert 50 * m[:, d].var() == pytest.approx(10 * shape * scale ** 2, rel=0.1), d


def test_gamma_mle_is_the_likelihood_maximum():
    rng = np.random.default_rng(5)
    for shape, scale in ((0.3, 2.0), (4.0, 0.1), (80.0, 0.004)):
        x = rng.gamma(shape, scale, 20_000)
        a, th = pf.fit_gamma_mle(x)
        a_ref, _, th_ref = st.gamma.fit(x, floc=0)
        assert a == pytest.approx(a_ref, rel=1e-5) and th == pytest.approx(th_ref, rel=1e-5)
        # stationarity of the log likelihood: log a - digamma(a) = log mean - mean log
        from scipy.special import digamma
        assert np.log(a) - digamma(a) == pytest.approx(np.log(x.mean()) - np.log(x).mean(), rel=1e-9)
        assert a * th == pytest.approx(x.mean(), rel=1e-12)


def test_fit_fallback_chain():
    rng = np.random.default_rng(6)
    good = rng.gamma(3.0, 0.01, 5000)
    with_zeros = good.copy()
    with_zeros[::7] = 0.0
    zeros = np.zeros(5000)
    with pytest.raises(pf.DomainError):
        pf.fit_gamma_mle(with_zeros)
    d = pf.FitInfectionDistributions(np.stack([zeros, with_zeros, good], axis=1), horizon=3, sample_size=50)
    assert d[0] == (1e-100, 1e-100)                                  # src/init.jl:23
    w = np.minimum(np.log(1.0 / np.where(with_zeros > 0, with_zeros, 1e-300)), 10.0)
    assert d[1][0] == 1 / 50 and d[1][1] == pytest.approx(50 * np.sum(w * with_zeros) / np.sum(w))   # :18-20
    a, th = pf.fit_gamma_mle(good)
    assert d[2] == (a / 50, th * 50)                                  # :13-14


def test_regenerated_tables_agree_with_builtin_defaults(tables):
    k, th = np.array(tables["Infdistributions"]).T
    assert (k[:2] == 1e-100).all() and (th[:2] == 1e-100).all()       # nobody is infectious before day 2.5
    assert k[2] == 1 / 50                                             # zeros in the column: Exponential branch
    np.testing.assert_allclose(k[3:], P.INF_SHAPE[3:], rtol=3e-2)
    np.testing.assert_allclose(th[2:], P.INF_SCALE[2:], rtol=3e-2)
    t6 = pf.generate_tables(seed=1, LOD=6)
    np.testing.assert_allclose(t6["pos_test_probs"], P.POS_TEST_PROBS[6], atol=6e-3)
    assert pf.generate_tables(seed=1, LOD=6)["pos_test_probs"] == t6["pos_test_probs"]        # seeded


def test_csv_round_trip_and_constructors(tmp_path, tables):
    inf_p, vl_p = str(tmp_path / "Sample50.csv"), str(tmp_path / "raw_viral_load.csv")
    pf.write_csvs(tables["infections"][:2000], tables["viral_loads"][:2000], inf_p, vl_p)
    assert open(inf_p).readline().strip() == ",".join(str(j) for j in range(21))
    t = pf.tables_from_csv(inf_p, vl_p, LOD=5)
    want = pf.FitInfectionDistributions(tables["infections"][:2000])
    assert t["Infdistributions"] == want                             # %.17g round-trips binary64
    pomdp = pf.CovidPOMDP_from_csv(inf_p, vl_p, LOD=5, test_delay=2, N=500_000)
    assert pomdp.horizon == 14 and pomdp.test_delay == 2 and pomdp.N == 500_000
    assert pomdp.Infdistributions == tuple(want)
    gen = pf.CovidPOMDP_generated(seed=0, n=20_000, LOD=4, horizon=18)     # a LOD / horizon the built-in tables lack
    assert gen.horizon == 18 and 0 < gen.pos_test_probs[4] <= 1
    assert gen.to_c().H == 18


def test_oracle_step_keeps_bounds_with_generated_tables():
    """The reference's own property (test/InfectionSimTest.jl:14-30) on a model
    built from generated tables."""
    from oracle import pyoracle
    pomdp = pf.CovidPOMDP_generated(seed=2, n=20_000, LOD=4)
    orc = pyoracle.Oracle(pomdp.to_c())
    st = orc.init_simplex(5, seed=77)
    for ai, a in enumerate((0.0, 0.3, 1.0)):
        out, hist, rew, fl = orc.rollout(st, 100, float(a), seed=5 + ai, step0=0)
        assert fl == 0
        assert (hist[:, :, :3] >= 0).all() and (hist[:, :, :3] <= pomdp.N).all()
        assert (hist[:, :, :3].sum(axis=2) == pomdp.N).all()
