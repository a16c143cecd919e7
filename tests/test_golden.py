"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the oracle): the oracle must keep reproducing them (CPU), and the HIP path
must reproduce them through the C ABI (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

import epidemicmodeling_amd as em
from epidemicmodeling_amd._lib import as_f64p, as_i64p, check
from oracle import pyoracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_reproduces_sampler_vectors():
    L = pyoracle.lib()
    d = np.load(os.path.join(G, "samplers.npz"))
    for j in range(d["n"].size):
        assert L.orc_binomial(99, j, 7, 5, 0, int(d["n"][j]), float(d["p"][j])) == d["binom_sympt"][j]
        assert L.orc_binomial(99, j, 7, 5, 1, int(d["n"][j]), float(d["p"][j])) == d["binom_pos"][j]
    for j in range(d["shape"].size):
        assert L.orc_gamma(4242, j, 3, 9, float(d["shape"][j]), float(d["scale"][j])) == d["gamma"][j]


@pytest.mark.parametrize("delay", [0, 1, 5])
def test_oracle_reproduces_step_vectors(delay):
    d = np.load(os.path.join(G, f"step_delay{delay}.npz"))
    orc = pyoracle.Oracle(em.make_pomdp(LOD=5, test_delay=delay).to_c())
    st = (d["S"], d["I"], d["R"], d["T"])
    out, ni, po, fl = orc.step_injected(st, d["a"], d["sympt"], d["pos"], d["gam"])
    for k, name in enumerate("SIRT"):
        assert (out[k] == d["inj_" + name]).all()
    assert (ni == d["inj_new"]).all() and fl == d["inj_flags"]
    out, o, r, ni, fl = orc.step_batch(st, d["a"], d["a"][::-1].copy(), seed=0x5EED, step=3)
    for k, name in enumerate("SIRT"):
        assert (out[k] == d["rng_" + name]).all()
    assert (o == d["rng_o"]).all() and (r == d["rng_r"]).all() and (ni == d["rng_new"]).all()


def test_oracle_reproduces_filter_vectors():
    d = np.load(os.path.join(G, "pf_update.npz"))
    orc = pyoracle.Oracle(em.make_pomdp(LOD=5, test_delay=1).to_c())
    post, info = orc.pf_update((d["S"], d["I"], d["R"], d["T"]), float(d["a"]), int(d["o"]), int(d["seed"]), int(d["step"]))
    for k, name in enumerate("SIRT"):
        assert (post[k] == d["post_" + name]).all()
    assert (info["anc"] == d["anc"]).all() and (info["lw"] == d["lw"]).all() and info["W"] == int(d["W"])


@pytest.mark.gpu
def test_device_reproduces_sampler_vectors():
    d = np.load(os.path.join(G, "samplers.npz"))
    ctx = em.interface.context_for(em.make_pomdp(LOD=5))
    n, p = np.ascontiguousarray(d["n"]), np.ascontiguousarray(d["p"])
    for which, key in ((0, "binom_sympt"), (1, "binom_pos")):
        out = np.empty(n.size, np.int64)
        check(ctx.lib.epi_sample_binomial(ctx.handle, n.size, as_i64p(n), as_f64p(p), 99, 7, 5, which, as_i64p(out)))
        assert (out == d[key]).all()
    shape, scale = np.ascontiguousarray(d["shape"]), np.ascontiguousarray(d["scale"])
    out = np.empty(shape.size)
    check(ctx.lib.epi_sample_gamma(ctx.handle, shape.size, as_f64p(shape), as_f64p(scale), 4242, 3, 9, as_f64p(out)))
    assert (out.view(np.uint64) == d["gamma"].view(np.uint64)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [0, 1, 5])
def test_device_reproduces_step_vectors(delay):
    d = np.load(os.path.join(G, f"step_delay{delay}.npz"))
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    n, H = d["S"].size, pomdp.horizon
    b = em.ParticleCollection(em.interface.context_for(pomdp), n)
    b.upload(d["S"], d["I"], d["R"], d["T"])
    a = np.ascontiguousarray(d["a"])
    ni = np.empty(n, np.int64); po = np.empty((n, H), np.int64); fl = C.c_uint32()
    check(b.ctx.lib.epi_step_injected(b.ctx.handle, b.handle, as_f64p(a), 1, as_i64p(np.ascontiguousarray(d["sympt"])),
                                      as_i64p(np.ascontiguousarray(d["pos"])), as_f64p(np.ascontiguousarray(d["gam"])),
                                      as_i64p(ni), as_i64p(po), C.byref(fl)))
    out = b.download()
    for k, name in enumerate("SIRT"):
        assert (out[k] == d["inj_" + name]).all()
    assert (ni == d["inj_new"]).all() and fl.value == d["inj_flags"]
    b.upload(d["S"], d["I"], d["R"], d["T"])
    o, r, fl2 = em.step_batch(b, a, a[::-1].copy(), seed=0x5EED, step=3)
    out = b.download()
    for k, name in enumerate("SIRT"):
        assert (out[k] == d["rng_" + name]).all()
    assert (o == d["rng_o"]).all() and (r.view(np.uint64) == d["rng_r"].view(np.uint64)).all() and fl2 == d["rng_flags"]


@pytest.mark.gpu
def test_device_reproduces_filter_vectors():
    d = np.load(os.path.join(G, "pf_update.npz"))
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    n = d["S"].size
    b = em.ParticleCollection(em.interface.context_for(pomdp), n)
    b.upload(d["S"], d["I"], d["R"], d["T"])
    upd = em.BootstrapFilter(pomdp, n, seed=int(d["seed"]), step=int(d["step"]))
    em.update(upd, b, em.Action(float(d["a"])), int(d["o"]))
    m = em.mean(b)
    assert (m.S, m.R) == (int(d["mean_S"]), int(d["mean_R"])) and (m.I == d["mean_I"]).all() and (m.Tests == d["mean_T"].T).all()
    out = b.download()
    for k, name in enumerate("SIRT"):
        assert (out[k] == d["post_" + name]).all()
    assert abs(upd.last_ess - float(d["ess"])) <= 1e-9 * float(d["ess"])
