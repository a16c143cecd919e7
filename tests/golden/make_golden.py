"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/epi_oracle.c).

    python tests/golden/make_golden.py

The vectors are outputs of THIS BUILD's restatement of the reference algorithm
(parity unpinned: the reference is Julia, cannot run where this is built, and
ships no vectors of its own).  They pin the oracle against drift and give the
GPU tests fixed inputs/outputs that do not depend on the oracle being rebuilt.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import epidemicmodeling_amd as em          # noqa: E402  (host-side parameter tables only)
from conftest import random_states         # noqa: E402
from oracle import pyoracle                # noqa: E402


def main():
    L = pyoracle.lib()
    rng = np.random.default_rng(20261111)

    # 1. samplers: one draw per element, element j on the streams of slot j
    n = np.array([0, 1, 3, 10, 19, 20, 21, 50, 100, 999, 1000, 5000, 10000, 123456, 1000000, 2 ** 31 - 1] * 16, dtype=np.int64)
    p = np.tile(np.array([0.0, 1e-6, 0.003, 0.05, 0.2, 0.4999, 0.5, 0.5001, 0.75, 0.93, 0.999, 1.0, 0.0524, 0.3, 0.0099, 0.6]), 16)
    p = np.roll(p, 5)[: n.size]
    bsym = np.array([L.orc_binomial(99, j, 7, 5, 0, int(n[j]), float(p[j])) for j in range(n.size)], dtype=np.int64)
    bpos = np.array([L.orc_binomial(99, j, 7, 5, 1, int(n[j]), float(p[j])) for j in range(n.size)], dtype=np.int64)
    shape = np.concatenate([[0.0, 1e-96, 1e-9, 0.02, 0.5, 0.999, 1.0, 1.5, 3.3], 10.0 ** rng.uniform(-3, 6, 247)])
    scale = 10.0 ** rng.uniform(-2, 1, shape.size)
    gam = np.array([L.orc_gamma(4242, j, 3, 9, float(shape[j]), float(scale[j])) for j in range(shape.size)])
    np.savez_compressed(os.path.join(HERE, "samplers.npz"), n=n, p=p, binom_sympt=bsym, binom_pos=bpos,
                        shape=shape, scale=scale, gamma=gam)

    # 2. injected step and Philox-driven step, test_delay 0 / 1 / 5
    for delay in (0, 1, 5):
        pomdp = em.make_pomdp(LOD=5, test_delay=delay)
        orc = pyoracle.Oracle(pomdp.to_c())
        m, H = 64, pomdp.horizon
        st = random_states(pomdp, m, rng)
        a = rng.random(m)
        a[:4] = [0.0, 1.0, 0.1, 0.3]
        S, I, R, T = st
        sympt = np.floor(rng.random((m, H)) * 0.3 * I).astype(np.int64)
        nt = np.floor((I - T.sum(axis=2)).astype(np.float64) * a[:, None]).astype(np.int64)
        pos = np.floor(rng.random((m, H)) * np.maximum(nt, 0)).astype(np.int64)
        g = rng.gamma(2.0, 50.0, size=(m, H)) * (rng.random((m, H)) > 0.2)
        out, ni, po, fl = orc.step_injected(st, a, sympt, pos, g)
        out2, o2, r2, ni2, fl2 = orc.step_batch(st, a, a[::-1].copy(), seed=0x5EED, step=3)
        np.savez_compressed(os.path.join(HERE, f"step_delay{delay}.npz"), S=S, I=I, R=R, T=T, a=a, sympt=sympt, pos=pos, gam=g,
                            inj_S=out[0], inj_I=out[1], inj_R=out[2], inj_T=out[3], inj_new=ni, inj_flags=fl,
                            rng_S=out2[0], rng_I=out2[1], rng_R=out2[2], rng_T=out2[3], rng_o=o2, rng_r=r2, rng_new=ni2, rng_flags=fl2)

    # 3. one complete filter update and the belief mean
    pomdp = em.make_pomdp(LOD=5, test_delay=1)
    orc = pyoracle.Oracle(pomdp.to_c())
    m = 768
    st = orc.init_normal(m, 10_000.0, 5_000.0, 0x5EED)
    st = orc.step_batch(st, 0.3, seed=1, step=0)[0]            # so that pending tests are present
    _, mu, _ = orc.log_weights(st, 0.3, 0)
    o = int(np.quantile(mu, 0.75))
    post, info = orc.pf_update(st, 0.3, o, 0x5EED, 9)
    mS, mI, mR, mT, _ = orc.mean(post)
    np.savez_compressed(os.path.join(HERE, "pf_update.npz"), S=st[0], I=st[1], R=st[2], T=st[3], a=0.3, o=o, seed=0x5EED, step=9,
                        post_S=post[0], post_I=post[1], post_R=post[2], post_T=post[3], anc=info["anc"], lw=info["lw"],
                        W=np.uint64(info["W"]), ess=info["ess"], mean_S=mS, mean_I=mI, mean_R=mR, mean_T=mT)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
