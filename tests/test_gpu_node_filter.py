"""The sharded update behind the C ABI (epi_mctx: one host thread, several shards) against the oracle and against the
one-device update.  The box has one GPU, so the shards share device 0 and the small exchanges go through host memory;
the RCCL leg runs with a single shard."""
import numpy as np
import pytest

import epidemicmodeling_amd as em
from epidemicmodeling_amd.distributed import NodeFilter
from oracle import pyoracle

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def _ll(info, n):
    F = pyoracle.lib().orc_frac_bits(n)
    return info["lwmax"] + np.log(info["W"] / 2.0 ** F / n) - 0.5 * np.log(2 * np.pi)


@pytest.mark.parametrize("delay,G", [(0, 1), (0, 2), (0, 5), (1, 3)])
def test_node_filter_bit_exact(delay, G):
    """Ragged shards, ordinary and far-out observations (the far one leaves nearly all weight in one shard, which
    then sends more particles than it holds: the piecewise send path), three updates: every posterior particle, the
    diagnostics and the belief mean equal the oracle's."""
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 20_011
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    f = NodeFilter(pomdp, n, [0] * G, seed=SEED)
    f.init_normal()
    for x, y in zip(f.download(), st):
        assert (x == y).all()
    _, mu, _ = orc.log_weights(st, 0.2, 0)
    moved = []
    for step, o in enumerate((int(np.median(mu)), int(2.5 * np.median(mu)), int(np.median(mu)) // 2)):
        got = f.update(0.2, o)
        st, info = orc.pf_update(st, 0.2, o, SEED, step)
        assert got["flags"] == info["flags"]
        assert abs(got["ess"] - info["ess"]) <= 1e-9 * info["ess"]
        assert abs(got["loglik"] - _ll(info, n)) <= 1e-9 * max(1.0, abs(_ll(info, n)))
        for name, x, y in zip("SIRT", f.download(), st):
            assert (x == y).all(), (name, step)
        moved.append(got["moved"])
        _, mu, _ = orc.log_weights(st, 0.2, 0)
    if G > 1:
        # the far-out observation leaves nearly all weight in one shard, which then sends (nearly) every slot the other
        # shards own: more than it holds itself once there are three shards or more
        assert max(moved) > (n // G if G > 2 else 0.95 * (n // G))
    else:
        assert moved == [0, 0, 0]
    eS, eI, eR, eT, bad = orc.mean(st)
    S, I, R, T = f.mean()
    assert (S, R) == (eS, eR) and (I == eI).all() and (T == eT.T).all() and not bad
    f.close()


@pytest.mark.parametrize("delay,G,overlap", [(0, 3, True), (0, 3, False), (1, 5, True), (0, 8, True)])
def test_node_filter_updates_back_to_back(delay, G, overlap):
    """Five updates with nothing read in between: from the second on, an update finds the blocks of its predecessor
    still in flight and steps around them (unless told not to); the posterior after the last equals the oracle's."""
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 50_021
    st = orc.init_normal(n, 10_000.0, 5_000.0, SEED)
    st[1][: n // 2] = st[1][: n // 2] * 15 // 16         # a tilt between the halves: particles change shard every update
    st[0][:] = pomdp.N - st[1].sum(axis=1) - st[2]
    f = NodeFilter(pomdp, n, [0] * G, seed=SEED, overlap=overlap)
    f.upload(*st)
    _, mu, _ = orc.log_weights(st, 0.2, 0)
    o = int(np.median(mu))
    moved = 0
    for step in range(5):
        got = f.update(0.2, o)
        st, info = orc.pf_update(st, 0.2, o, SEED, step)
        assert got["flags"] == info["flags"] and abs(got["ess"] - info["ess"]) <= 1e-9 * info["ess"]
        moved += got["moved"]
    assert moved > 0
    assert (f.overlapped_updates > 0) == overlap
    for name, x, y in zip("SIRT", f.download(), st):
        assert (x == y).all(), name
    # and on from there, after the population has been read
    got = f.update(0.2, o)
    st, info = orc.pf_update(st, 0.2, o, SEED, 5)
    eS, eI, eR, eT, bad = orc.mean(st)
    S, I, R, T = f.mean()
    assert (S, R) == (eS, eR) and (I == eI).all() and (T == eT.T).all() and not bad
    f.close()


def _random_shapes_and_observations():
    rng = np.random.default_rng(20261111)
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    for case in range(24):
        G = int(rng.integers(1, 9))
        n = int(rng.integers(max(G, 50), 6000))
        seed = int(rng.integers(1, 2 ** 31))
        st = orc.init_normal(n, 10_000.0, 5_000.0, seed)
        tilt = rng.random()
        st[1][: n // 2] = (st[1][: n // 2] * (0.8 + 0.2 * tilt)).astype(np.int64)
        st[0][:] = pomdp.N - st[1].sum(axis=1) - st[2]
        f = NodeFilter(pomdp, n, [0] * G, seed=seed, overlap=bool(case % 4))
        f.upload(*st)
        for step in range(5):
            a = 0.0 if rng.random() < 0.15 else float(rng.uniform(0.05, 0.6))
            _, mu, _ = orc.log_weights(st, a, 0)
            kind = rng.random()
            o = int(np.quantile(mu, rng.random())) if kind < 0.7 else int(mu.max() * rng.uniform(1.5, 4.0)) + 3
            got = f.update(a, o)
            st, info = orc.pf_update(st, a, o, seed, step)
            assert got["flags"] == info["flags"], (case, step)
            assert abs(got["ess"] - info["ess"]) <= 1e-9 * max(info["ess"], 1.0), (case, step)
        for name, x, y in zip("SIRT", f.download(), st):
            assert (x == y).all(), (case, name)
        f.close()


def test_node_filter_random_shapes_and_observations():
    """Small random populations over 1 to 8 shards (some shards smaller than a workgroup), five updates each with
    observations from typical to impossible and now and then no testing at all (every weight zero), nothing read in
    between: collapses onto one shard, exchanges left in flight and updates that keep every particle follow each other in
    every order.  Posterior and diagnostics equal the oracle's."""
    _random_shapes_and_observations()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_node_filter_random_shapes_through_the_branches_several_devices_take(testing_lib, mode):
    """The same cases with the shards of the one device treated as if they lay on different devices: particle blocks
    go through hipMemcpyPeerAsync (mode bit 0; receivers wait for the senders' events as always), and the global
    maximum goes through host memory as between devices without access to each other (bit 1) instead of being read from
    the other shards' memory."""
    assert testing_lib.epi_debug_node_exchange(mode) == 0
    try:
        _random_shapes_and_observations()
    finally:
        testing_lib.epi_debug_node_exchange(0)


def test_node_filter_all_weights_zero_keeps_the_particles():
    pomdp = em.make_pomdp(LOD=5)
    orc = pyoracle.Oracle(pomdp.to_c())
    n = 4_000
    st = orc.init_normal(n, 10_000.0, 5_000.0, 3)
    f = NodeFilter(pomdp, n, [0, 0, 0], seed=3)
    f.init_normal()
    got = f.update(0.0, 7)                               # a = 0: nobody tested, o = 7 impossible
    st, info = orc.pf_update(st, 0.0, 7, 3, 0)
    assert got["flags"] == info["flags"] and got["flags"] & 8 and got["moved"] == 0
    for x, y in zip(f.download(), st):
        assert (x == y).all()
    f.close()


def test_node_filter_over_rccl_with_one_shard():
    """use_rccl = 1: librccl.so is loaded at run time, a communicator of one device is created and the all-reduce and
    all-gather of an update run on it; the posterior equals the one-device update's."""
    pomdp = em.make_pomdp(LOD=5)
    n = 30_000
    f = NodeFilter(pomdp, n, [0], seed=SEED, rccl=True)
    f.init_normal()
    b = em.initialbelief(pomdp, n, seed=SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=SEED)
    for o in (9000, 8000):
        got = f.update(0.2, o)
        em.update(upd, b, em.Action(0.2), o)
        assert (got["loglik"], got["ess"], got["flags"]) == (upd.last_loglik, upd.last_ess, upd.last_flags)
    for x, y in zip(f.download(), b.download()):
        assert (x == y).all()
    f.close(); b.close()


def test_node_filter_rejects_what_it_cannot_run():
    from epidemicmodeling_amd._lib import EpiError
    pomdp = em.make_pomdp(LOD=5)
    with pytest.raises(EpiError, match="one device per shard"):
        NodeFilter(pomdp, 1000, [0, 0], rccl=True)
    with pytest.raises(EpiError, match="fewer particles than shards"):
        NodeFilter(pomdp, 2, [0, 0, 0])
    with pytest.raises(EpiError, match="device"):
        NodeFilter(pomdp, 1000, [0, 99])
