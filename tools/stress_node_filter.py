"""Randomised comparison of the in-library sharded update (epi_mctx) with the oracle at sizes above the test suite's:
10 + 10 populations of 20 to 120 thousand particles over 2 to 8 shards, 8 updates each, nothing read in between."""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import epidemicmodeling_amd as em
from epidemicmodeling_amd.distributed import NodeFilter
from oracle import pyoracle
rng = np.random.default_rng(7)
for delay in (0, 2):
    pomdp = em.make_pomdp(LOD=5, test_delay=delay)
    orc = pyoracle.Oracle(pomdp.to_c())
    for case in range(10):
        G = int(rng.integers(2, 9)); n = int(rng.integers(20000, 120000)); seed = int(rng.integers(1, 2**31))
        st = orc.init_normal(n, 10_000.0, 5_000.0, seed)
        st[1][: n // 2] = (st[1][: n // 2] * (0.85 + 0.15 * rng.random())).astype(np.int64)
        st[0][:] = pomdp.N - st[1].sum(axis=1) - st[2]
        f = NodeFilter(pomdp, n, [0] * G, seed=seed)
        f.upload(*st)
        for step in range(8):
            a = float(rng.uniform(0.05, 0.6))
            _, mu, _ = orc.log_weights(st, a, 0)
            o = int(np.quantile(mu, rng.random())) if rng.random() < 0.8 else int(mu.max() * 2) + 3
            got = f.update(a, o)
            st, info = orc.pf_update(st, a, o, seed, step)
            assert got["flags"] == info["flags"]
        for x, y in zip(f.download(), st):
            assert (x == y).all(), (delay, case)
        print(delay, case, G, n, "ok, overlapped updates", f.overlapped_updates, flush=True)
        f.close()
