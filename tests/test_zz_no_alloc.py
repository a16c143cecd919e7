"""epi_pf_update_noalloc (the update with its scratch in a file-scope device array) against epi_pf_update."""
import ctypes as C

import numpy as np
import pytest

import epidemicmodeling_amd as em
from epidemicmodeling_amd._lib import EpiError, check

pytestmark = pytest.mark.gpu

SEED = 0xA110C


def _update(lib, name, ctx, b, a, o, step):
    ll, ess, fl = C.c_double(), C.c_double(), C.c_uint32()
    check(getattr(lib, name)(ctx.handle, b.handle, a, o, SEED, step, C.byref(ll), C.byref(ess), C.byref(fl)))
    return ll.value, ess.value, fl.value


def _pair(pomdp, n, seed):
    """the same prior twice"""
    b0 = em.initialbelief(pomdp, n, seed=seed)
    return b0, b0.copy()


def _observation(pomdp, b, a):
    S, I, R, T = b.download()
    tested = np.floor((I - T.sum(axis=2)) * a)
    return int(np.median((tested * np.asarray(pomdp.pos_test_probs)).sum(axis=1)))


def test_noalloc_update_equals_the_allocating_one_on_two_calls_in_a_row():
    pomdp = em.make_pomdp(LOD=5)
    a = 0.2
    # first a larger population, then a smaller one: the second call finds the first one's tile totals, range list
    # and scalars in the shared array
    for n, step in ((50_000, 0), (3_000, 1)):
        ref, got = _pair(pomdp, n, SEED + n)
        ctx, lib = ref.ctx, ref.ctx.lib
        o = _observation(pomdp, ref, a)
        for t in range(2):                                   # the second round also reads through the ancestor index
            want = _update(lib, "epi_pf_update", ctx, ref, a, o, step + 10 * t)
            have = _update(lib, "epi_pf_update_noalloc", ctx, got, a, o, step + 10 * t)
            assert have == want                              # log-likelihood, ESS, flags: the same bits
            for x, y in zip(ref.download(), got.download()):
                assert (x == y).all()
        ref.close(); got.close()


def test_noalloc_update_with_collapsed_weights_lists_long_ranges():
    # an observation far in the tail: a handful of particles take all the children (ranges beyond 8192 slots go
    # through the range list, which lives in the static array too)
    pomdp = em.make_pomdp(LOD=5)
    ref, got = _pair(pomdp, 40_000, SEED)
    ctx, lib = ref.ctx, ref.ctx.lib
    o = 3 * _observation(pomdp, ref, 0.2)
    want = _update(lib, "epi_pf_update", ctx, ref, 0.2, o, 0)
    have = _update(lib, "epi_pf_update_noalloc", ctx, got, 0.2, o, 0)
    assert have == want and want[1] < 50                     # ESS: collapsed
    for x, y in zip(ref.download(), got.download()):
        assert (x == y).all()
    ref.close(); got.close()


def test_noalloc_update_rejects_what_it_cannot_hold():
    pomdp = em.make_pomdp(LOD=5)
    big = em.ParticleCollection(em.interface.context_for(pomdp), (1 << 27) + 1)
    with pytest.raises(EpiError, match="exceed the static scratch's maximum") as e:
        _update(big.ctx.lib, "epi_pf_update_noalloc", big.ctx, big, 0.2, 100, 0)
    assert e.value.code == -4
    big.close()
    shard = em.ParticleCollection(em.interface.context_for(pomdp), 1000)
    check(shard.ctx.lib.epi_belief_set_shard(shard.handle, 4000, 1000))
    with pytest.raises(EpiError, match="shard"):
        _update(shard.ctx.lib, "epi_pf_update_noalloc", shard.ctx, shard, 0.2, 100, 0)
    shard.close()
