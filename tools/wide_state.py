"""Rollout rate against the state width H for the default choice of step kernel (cooperative with one thread per particle
up to H = 17, two up to H = 28, four beyond) and for the per-thread kernel at every width.  GPU box,
repository root: python3 tools/wide_state.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import epidemicmodeling_amd as em
from epidemicmodeling_amd import _lib
# the choice of step kernel is a switch of the testing build (include/epi_testing.h): make it the library every context uses
lib = _lib._lib = _lib.load_testing()
for H in (14, 16, 21, 28, 42, 56, 64):
    pomdp = em.make_pomdp(LOD=5, horizon=H)
    n, T = 4_000_000, 10
    for impl in (1, 0):
        lib.epi_debug_step_impl(impl)
        b = em.random_states(pomdp, n, seed=1)
        em.rollout_batch(b, 2, 0.2, seed=3, want_hist=False, want_rew=False)
        t0 = time.perf_counter()
        em.rollout_batch(b, T, 0.2, seed=2, want_hist=False, want_rew=False)
        dt = time.perf_counter() - t0
        print(f"H={H} impl={('per-thread', 'default')[impl]}: {n*T/dt:.3e} particle-steps/s")
        b.close()
lib.epi_debug_step_impl(1)
