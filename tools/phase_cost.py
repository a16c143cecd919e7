#!/usr/bin/env python3
"""Time of the fused step kernel with parts of the model switched off through the parameters
(no gammas: shape 0; no binomials: q = p = 0), to see where the kernel time goes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epidemicmodeling_amd as em
from epidemicmodeling_amd._lib import check

def run(name, pomdp, n=10_000_000, a=0.2, steps=5, impl=1):
    b = em.initialbelief(pomdp, n, seed=1)
    ctx, lib = b.ctx, b.ctx.lib
    lib.epi_debug_step_impl(impl)
    for t in range(2):
        check(lib.epi_pf_update_async(ctx.handle, b.handle, a, 9000, 1, t))
    ll, ess, fl = C.c_double(), C.c_double(), C.c_uint32()
    check(lib.epi_pf_collect(ctx.handle, b.handle, C.byref(ll), C.byref(ess), C.byref(fl)))
    check(lib.epi_ctx_timing(ctx.handle, 1))
    for t in range(2, 2 + steps):
        check(lib.epi_pf_update_async(ctx.handle, b.handle, a, 9000, 1, t))
    s, c = C.c_double(), C.c_int64()
    check(lib.epi_ctx_timing_read(ctx.handle, C.byref(s), C.byref(c)))
    check(lib.epi_ctx_timing(ctx.handle, 0))
    print(f"{name:42s} impl={impl} k_step {s.value / c.value:8.3f} ms per 1e7 particles")
    b.close()

full = em.make_pomdp()
zero_g = [(0.0, 1.0)] * 14
variants = [
    ("full", full),
    ("no gammas", em.make_pomdp(Infdistributions=zero_g)),
    ("no binomials (q=p=0)", em.make_pomdp(symptomatic_isolation_prob=0.0, pos_test_probs=[0.0] * 14)),
    ("no isolation draws (q=0)", em.make_pomdp(symptomatic_isolation_prob=0.0)),
    ("no test draws (p=0)", em.make_pomdp(pos_test_probs=[0.0] * 14)),
    ("neither", em.make_pomdp(symptomatic_isolation_prob=0.0, pos_test_probs=[0.0] * 14, Infdistributions=zero_g)),
]
for impl in ([int(sys.argv[1])] if len(sys.argv) > 1 else (1, 0)):
    for name, p in variants:
        run(name, p, impl=impl)
