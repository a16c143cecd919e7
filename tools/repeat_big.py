#!/usr/bin/env python3
"""The benchmark's population filtered twice from the same seeds (developer tool; GPU box, repository root):
    python3 tools/repeat_big.py [particles, default 1e8]
Eight updates each; log likelihood, effective sample size, flags and the integer column sums behind the posterior mean must
agree bit for bit between the two runs (a race between the phases of the step kernel shows as a difference here long before
any statistical test sees it)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import epidemicmodeling_amd as em
import bench
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
pomdp = em.make_pomdp()
obs = bench.observations(pomdp, 8)
res = []
for rep in range(2):
    b = em.initialbelief(pomdp, n, seed=bench.SEED)
    upd = em.BootstrapFilter(pomdp, n, seed=bench.SEED)
    out = []
    for t in range(8):
        em.update(upd, b, em.Action(bench.ACTION), obs[t])
        m = em.mean(b, pomdp)
        out.append((upd.last_loglik, upd.last_ess, upd.last_flags, int(m.S), tuple(int(x) for x in m.I), int(m.R)))
    res.append(out)
    b.close()
    print("rep", rep, out[-1][:3], flush=True)
ok = res[0] == res[1]
print("identical" if ok else "DIFFERENT")
for t, (x, y) in enumerate(zip(*res)):
    if x != y: print(t, x, y)
sys.exit(0 if ok else 1)
