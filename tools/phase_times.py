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
import bench
pomdp = em.make_pomdp()
b = em.initialbelief(pomdp, n, seed=bench.SEED)
upd = em.BootstrapFilter(pomdp, n, seed=bench.SEED)
obs = bench.observations(pomdp, 6)
L = _lib.load()
out = (C.c_ulonglong * 32)()
L.epi_debug_phase_times.argtypes = [C.POINTER(C.c_ulonglong)]
for step in range(6):
    em.update(upd, b, em.Action(bench.ACTION), obs[step])
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
print(f"rejection draws in phase B per workgroup-step {out[7] / groups:.1f}, of them past the squeeze (series bounds evaluated) {out[6] / groups:.1f} = {100.0 * out[6] / max(out[7], 1):.1f} %")
wd = max(out[21], 1)
print(f"phase B, (wave, draw) pairs of rejection-or-inversion bins per workgroup-step {out[21] / groups:.1f}: with a rejection lane {100.0 * out[16] / wd:.1f} %, "
      f"with an inversion lane {100.0 * out[17] / wd:.1f} %, with both {100.0 * out[18] / wd:.1f} %; inversion lanes per pair {out[19] / wd:.1f}")
print(f"inversion trips (eight terms, wave level) per workgroup-step {out[20] / groups:.1f} = {out[20] / max(out[17], 1):.2f} per pair with an inversion lane; live lanes per trip {out[22] / max(out[20], 1):.1f}")
