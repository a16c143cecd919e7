#!/usr/bin/env python3
"""Stack traffic inside the kernels themselves: every scratch load / store of a kernel's own body with the loop nest it
sits in (a spill outside the loops costs its bytes once per thread and launch; one inside a loop, once per trip).
    tools/kernel_spills.py [name filter, default k_step_coop] [extra compiler flags]"""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "epidemicmodeling_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else "k_step_coop"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "--cuda-device-only", "-S"] + sys.argv[2:] + \
      ["-o", "/tmp/epi_spills.s", "epi_gfx950.hip"]
subprocess.run(cmd, cwd=here, check=True, capture_output=True)
txt = open("/tmp/epi_spills.s").read().splitlines()
starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
for i0 in starts:
    name = txt[i0].split(":")[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt not in dem or "(" not in dem:
        continue
    i1 = next(i for i in range(i0, len(txt)) if txt[i].startswith(".Lfunc_end"))
    body = txt[i0:i1]
    depth_at, depth = [], 0
    loops = {}                               # label -> depth from the comments
    ops = []
    cur = 0
    for i, l in enumerate(body):
        m = re.search(r"Loop Header: Depth=(\d+)", l)
        if m:
            cur = int(m.group(1))
        if re.match(r"^\.LBB\d+_\d+:", l) and "Loop" not in l and "Parent Loop" not in body[min(i + 1, len(body) - 1)]:
            pass
        if "scratch_" in l:
            ops.append((i, l.strip().split(";")[0].strip()))
    st = sum(1 for _, o in ops if "store" in o)
    ld = len(ops) - st
    print(f"{re.sub(r'[(]StepArgs.*', '', dem)}: {len(body)} lines, {st} scratch stores, {ld} loads")
    for i, o in ops:
        print(f"    line {i:5d}  {o}")
