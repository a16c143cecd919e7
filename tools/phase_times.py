#!/usr/bin/env python3
"""Clock ticks thread 0 of every workgroup of the cooperative step kernel spends per phase (developer tool).

Needs a variant build with the stamps compiled in:
    make -C epidemicmodeling_amd/csrc variant NAME=stamps EXTRA=-DEPI_PHASE_STAMPS
    EPI_GFX950_LIB=epidemicmodeling_amd/csrc/variants/libepi_stamps.so python3 tools/phase_times.py [particles]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import epidemicmodeling_amd as em
from epidemicmodeling_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pomdp = em.make_pomdp(LOD=5)
b = em.initialbelief(pomdp, n, seed=24301)
upd = em.BootstrapFilter(pomdp, n, seed=24301)
L = _lib.load()
out = (C.c_ulonglong * 16)()
L.epi_debug_phase_times.argtypes = [C.POINTER(C.c_ulonglong)]
o = 400
for step in range(6):
    em.update(upd, b, em.Action(0.2), o)
    if step == 2:
        L.epi_debug_phase_times(out)          # clear after the warm-up
L.epi_debug_phase_times(out)
names = ["A load/weights", "B first attempts", "C binomial rounds", "D bookkeeping + gamma", "E gamma rounds", "F outputs"]
tot = sum(out[:6])
groups = 3 * ((n + 255) // 256)
for k, nm in enumerate(names):
    print(f"{nm:24s} {out[k] / groups:10.0f} ticks per workgroup-step  {100.0 * out[k] / tot:5.1f} %")
print(f"{'total':24s} {tot / groups:10.0f}")
print(f"binomial rounds per workgroup-step {out[8] / groups:.2f}, queued draws {out[9] / groups:.1f} (round 1: {out[10] / groups:.1f}, of them rejected in the first phase {out[15] / groups:.1f}; round 2: {out[11] / groups:.1f})")
print(f"gamma rounds per workgroup-step {out[12] / groups:.2f}, queued terms {out[13] / groups:.1f} (round 1: {out[14] / groups:.1f})")
