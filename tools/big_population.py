#!/usr/bin/env python3
"""A population whose state rows pass 2^31 elements (3·10^8 particles, 42 GB with both buffers), checked through
properties that do not need a CPU run of that size:

  * the column sums of the population after two plain steps in one launch each equal the sum of the column sums of
    the same slots stepped as six shards of 5·10^7 (counter-based streams: a particle's draws depend on its slot
    number only, so any slip in 64-bit indexing shows as a difference);
  * three filter updates on the whole population: no flags, effective sample size in range, population conserved.

    python3 tools/big_population.py [particles] [chunk]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import epidemicmodeling_amd as em
from epidemicmodeling_amd._lib import check

SEED = 0x5EED
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
chunk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
assert n % chunk == 0

pomdp = em.make_pomdp()
ctx = em.interface.context_for(pomdp)
lib = ctx.lib
H, D = pomdp.horizon, pomdp.test_delay
nsums = 2 + H + (D + 1) * H


def sums(b):
    arr = (C.c_int64 * nsums)()
    check(lib.epi_belief_sums(ctx.handle, b.handle, arr))
    return np.array(arr, dtype=np.int64)


def two_steps(b):
    a, ap, fl = C.c_double(0.2), C.c_double(0.0), C.c_uint32()
    for step in range(2):
        check(lib.epi_step_batch(ctx.handle, b.handle, C.byref(a), 0, C.byref(ap), 0, SEED, step, None, None, None, C.byref(fl)))
        assert fl.value == 0, fl.value


t0 = time.perf_counter()
big = em.ParticleCollection(ctx, n)
check(lib.epi_belief_init_normal(big.handle, 10_000.0, 5_000.0, SEED))
s_init = sums(big)
two_steps(big)
s_big = sums(big)
print(f"whole population: {n} particles, init + 2 steps + sums in {time.perf_counter() - t0:.2f} s", flush=True)

acc_init = np.zeros(nsums, np.int64)
acc = np.zeros(nsums, np.int64)
for k in range(n // chunk):
    b = em.ParticleCollection(ctx, chunk)
    check(lib.epi_belief_set_shard(b.handle, n, k * chunk))
    check(lib.epi_belief_init_normal(b.handle, 10_000.0, 5_000.0, SEED))
    acc_init += sums(b)
    two_steps(b)
    acc += sums(b)
    b.close()
assert (acc_init == s_init).all(), (acc_init, s_init)
assert (acc == s_big).all(), (acc, s_big)
assert s_big[0] + s_big[1] + s_big[2:2 + H].sum() == pomdp.N * n          # S + R + sum I
print("column sums of the whole population equal the sums over its shards:", s_big[:4], "...", flush=True)

upd = em.BootstrapFilter(pomdp, n, seed=SEED)
for day in range(3):
    t0 = time.perf_counter()
    em.update(upd, big, em.Action(0.2), 9000 - 200 * day)
    assert upd.last_flags == 0 and 1 < upd.last_ess <= n
    print(f"update {day}: ess {upd.last_ess:.4g}, loglik {upd.last_loglik:.6f}, {time.perf_counter() - t0:.3f} s", flush=True)
s = sums(big)
assert s[0] + s[1] + s[2:2 + H].sum() == pomdp.N * n
big.close()
print("population conserved through three updates; ok")
