#!/usr/bin/env python3
"""Ensemble generation and ODE-loss evaluation at the size of the ODE-fit workload (10^7 trajectories, T = 30):
wall times and the loss kernel's HBM rate (it reads S, sum I, R of every trajectory-day: 12 bytes)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import epidemicmodeling_amd as em

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H = int(sys.argv[3]) if len(sys.argv) > 3 else 14
pomdp = em.make_pomdp(LOD=5, horizon=H)
b = em.random_states(pomdp, n, seed=1)
acts = np.random.default_rng(0).random(n)
t0 = time.perf_counter()
ens = em.Ensemble(b, T, acts, seed=2)
t1 = time.perf_counter()
print(f"ensemble: {n} trajectories x {T} days (H={H}) in {t1 - t0:.3f} s = {n * T / (t1 - t0):.3e} particle-steps/s, flags {ens.flags}")
for kind, p in (("SIR", [0.1, 0.3]), ("SEIR_CTRL", [0.1, 0.4, 0.7, 0.3, 0.5])):
    for sub in (1, 8):
        ens.loss(kind, p, substeps=sub)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            v = ens.loss(kind, p, substeps=sub)
        dt = (time.perf_counter() - t0) / reps
        print(f"loss {kind:9s} substeps {sub}: {dt * 1e3:8.3f} ms per evaluation, {12.0 * n * T / dt / 1e9:8.1f} GB/s of history read, loss {v:.6g}")
ens.close()
