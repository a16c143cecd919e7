#!/usr/bin/env python3
"""Wall times of the other benchmark configurations at full size on one GPU (the headline metric is bench.py's):
  closed loop   10^6-particle belief, 50-day closed loop with a fixed action, per-day belief summaries on the device
  scoring       10^5 action sequences x 64 rollouts each (6.4 10^6 trajectories), T = 20, discounted reward per sequence
  ensembles     10^7 trajectories x 30 days for H = 14, 28, 56 (random start states, one random action each)
Run from the repository root; every case runs twice and the second run is reported."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import epidemicmodeling_amd as em

SEED = 0x5EED


def closed_loop():
    pomdp = em.make_pomdp()
    n, T = 1_000_000, 50
    for rep in range(2):
        true = em.State_from((10_000.0, 5_000.0), pomdp, np.random.default_rng(2))
        b = em.initialbelief(pomdp, n, seed=SEED)
        t0 = time.perf_counter()
        em.SimulateClosedLoop(T, true, b, pomdp, lambda bel: em.Action(0.2), seed=SEED, keep_beliefs="summary")
        dt = time.perf_counter() - t0
        b.close()
    print(f"closed loop: {n} particles x {T} days in {dt:.3f} s = {n * T / dt:.3e} particle-steps/s "
          f"({dt / T * 1e3:.2f} ms per day: true step, update, mean and three order statistics of the belief)")


def scoring():
    pomdp = em.make_pomdp()
    b = em.initialbelief(pomdp, 10_000, seed=SEED)
    n_seq, R, T = 100_000, 64, 20
    seqs = np.random.default_rng(0).integers(0, 11, size=(n_seq, T)) / 10.0
    for rep in range(2):
        t0 = time.perf_counter()
        em.rollout_score(b, seqs, R, seed=7)
        dt = time.perf_counter() - t0
    b.close()
    print(f"scoring: {n_seq} sequences x {R} rollouts x {T} days in {dt:.3f} s = {n_seq * R * T / dt:.3e} particle-steps/s")


def ensembles(widths=(14, 28, 56)):
    for H in widths:
        pomdp = em.make_pomdp(LOD=5, horizon=H)
        n, T = 10_000_000, 30
        acts = np.random.default_rng(H).random(n)
        for rep in range(2):
            b = em.random_states(pomdp, n, seed=H)
            t0 = time.perf_counter()
            ens = em.Ensemble(b, T, acts, seed=H)
            dt = time.perf_counter() - t0
            ens.close(); b.close()
        print(f"ensemble H={H}: {n} trajectories x {T} days in {dt:.3f} s = {n * T / dt:.3e} particle-steps/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["closed_loop", "scoring", "ensembles"]
    for name in which:
        if name.startswith("ensembles:"):               # ensembles:28 -- one state width
            ensembles((int(name.split(":")[1]),))
        else:
            {"closed_loop": closed_loop, "scoring": scoring, "ensembles": ensembles}[name]()
