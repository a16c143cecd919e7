"""Host-side mirror of src/typedef.jl: parameter tables and constructors."""
import numpy as np
import pytest

from epidemicmodeling_amd import (Action, Array, State, State_from, State_random, make_pomdp, observation, population,
                                  unity_test_period)


def test_isolation_probs_follow_lognormal_cdf():
    # F(i) - F(i-1), F = cdf(LogNormal(1.644, 0.363)) (src/simulate.jl:56, src/typedef.jl:100)
    expect = [3e-6, .004401, .062087, .172382, .2232, .195942, .139199, .087632, .051404, .028933, .015909, .008644, .004675, .002529]
    p = make_pomdp(symptomatic_isolation_prob=1.0, asymptomatic_prob=0.0)
    assert np.allclose(p.isolation_probs(), expect, atol=1.5e-6)
    p2 = make_pomdp(symptomatic_isolation_prob=0.95, asymptomatic_prob=0.40)
    assert np.allclose(p2.isolation_probs(), np.array(expect) * 0.6 * 0.95, atol=1e-6)
    assert (make_pomdp(symptomatic_isolation_prob=0.0).isolation_probs() == 0).all()


def test_constructor_defaults_and_errors():
    p = make_pomdp()
    assert (p.test_delay, p.N, p.discount, p.inf_loss, p.test_loss, p.testrate_loss, p.test_period) == (0, 1_000_000, 0.95, 50.0, 1.0, 0.1, 1)
    assert p.horizon == 14 and len(p.pos_test_probs) == 14
    # the LOD = 5 table is the one the reference prints (notebooks/InfectionSimDemo.ipynb)
    assert make_pomdp(LOD=5).pos_test_probs[4] == 0.75496
    with pytest.raises(ValueError):
        make_pomdp(LOD=4)
    assert unity_test_period(make_pomdp(test_period=7)).test_period == 1
    c = make_pomdp(test_delay=3).to_c()
    assert (c.H, c.D, c.N) == (14, 3, 1_000_000) and c.k[2] == 0.02
    wide = make_pomdp(horizon=28)
    assert wide.horizon == 28 and wide.pos_test_probs[27] < wide.pos_test_probs[13]


def test_state_constructors():
    p = make_pomdp(test_delay=2)
    s = State_from(100, p)
    assert s.I[0] == 100 and s.S == p.N - 100 and s.R == 0 and s.Tests.shape == (3, 14) and s.prev_action == Action(0.0)
    s = State_from([5] * 14, p)
    assert population(s) == p.N and list(Array(s)) == [p.N - 70, 70, 0]
    s = State_from((10_000.0, 5_000.0), p, np.random.default_rng(0))
    assert (s.I >= 0).all() and population(s) == p.N
    with pytest.raises(ValueError):
        State_from("x", p)
    s = State_random(p, np.random.default_rng(1))
    assert population(s) == p.N and s.S >= 0 and s.R >= 0 and (s.I >= 0).all()


def test_observation_moments():
    p = make_pomdp(LOD=5)
    s = State_from([1000] * 14, p)
    m, sd = observation(p, s, Action(0.5))
    probs = np.array(p.pos_test_probs)
    assert np.isclose(m, (500 * probs).sum()) and np.isclose(sd ** 2, (500 * probs * (1 - probs)).sum())


def test_committed_bench_line_carries_the_contract_fields():
    """The bench line kept under profiles/ (written by bench.py on the GPU box) has the fields the measurement
    contract names, and its derived figures are consistent with each other."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r1_final", "bench.json")
    d = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "particle-steps/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and "workload" in d["config"]
    n, r = d["config"]["particles"], d["roofline"]
    assert np.isclose(d["value"], n / (d["ms_per_step"] * 1e-3), rtol=1e-9)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert np.isclose(r["achieved"], r["alg_bytes_per_particle_step"] * n / (r["kernel_ms"] * 1e-3) / 1e9, rtol=1e-9)
    assert np.isclose(r["frac"], r["achieved"] / r["peak"]) and r["traffic"] > r["alg_bytes_per_particle_step"] * n * 0.9
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
