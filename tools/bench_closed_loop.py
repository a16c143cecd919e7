#!/usr/bin/env python3
"""Config 2: the 10^6-particle closed loop over 50 days with a fixed action, days per device call 1 (one host wait per
day) and 50 (one per run); with and without the per-day belief summaries.  Second run of each reported."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import epidemicmodeling_amd as em

SEED = 0x5EED
pomdp = em.make_pomdp()
n, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 50
for period, keep in ((1, "summary"), (1, False), (50, "summary"), (50, False)):
    model = em.make_pomdp(test_period=period)
    for rep in range(2):
        true = em.State_from((10_000.0, 5_000.0), model, np.random.default_rng(2))
        b = em.initialbelief(model, n, seed=SEED)
        t0 = time.perf_counter()
        em.SimulateClosedLoop(T, true, b, model, lambda bel: em.Action(0.2), seed=SEED, keep_beliefs=keep)
        dt = time.perf_counter() - t0
        b.close()
    print(f"closed loop, {period:2d} day(s) per call, summaries {keep!s:7s}: {n} particles x {T} days in {dt * 1e3:.1f} ms = "
          f"{dt / T * 1e3:.3f} ms per day = {n * T / dt:.3e} particle-steps/s", flush=True)
