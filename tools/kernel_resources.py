#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel of the library (compiler remarks), one line per kernel:
    tools/kernel_resources.py [extra compiler flags] | grep k_step_coop"""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "epidemicmodeling_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950"] + sys.argv[1:] + \
      ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/epi_resources.so", "epi_gfx950.hip"]
out = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
cur, rows = None, {}
for ln in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass-analysis", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    r = rows[n]
    d = re.sub(r"\(StepArgs.*", "", d)
    print("%-72s VGPR %s AGPR %s SGPR %s spill V/S %s/%s scratch %s occ %s" % (
        d, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPR Spill"), r.get("SGPR Spill"),
        r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
