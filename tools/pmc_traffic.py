#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into bytes per particle per kernel.

rocprofv3 reports both counters in KiB... per dispatch.  On gfx950 FETCH_SIZE tallies 128-byte read requests of wide
streaming loads at 64 bytes; the kernels here load 4 or 8 bytes per lane, a width for which that factor is not
established, so the factor is calibrated on dispatches whose byte count is known exactly:
  k_gather        (bench.py --calibrate) reads the ancestor index and every row, 4 bytes per lane, and writes
                  every row: 4*W + 4 bytes read and 4*W bytes written per particle, the step kernel's own
                  access pattern
The corrected figure for the step kernel is  raw_fetch * fetch_factor + raw_write * write_factor.
"""
import csv, glob, json, sys, collections

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = sys.argv[3] if len(sys.argv) > 3 else None


def load(pat, counter):
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob(pat)):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0][:48]].append(float(row["Counter_Value"]))
    return acc


fetch = load("gpurun_out/pmc_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
write = load("gpurun_out/pmc_write/*/*_counter_collection.csv", "WRITE_SIZE")
KB = 1024.0
rep = {}
for k in sorted(set(fetch) | set(write)):
    f = [v * KB / n for v in fetch.get(k, [])]
    w = [v * KB / n for v in write.get(k, [])]
    rep[k] = {"fetch_B_per_particle": [round(x, 3) for x in f], "write_B_per_particle": [round(x, 3) for x in w]}
    print(f"{k:50s} fetch/particle {['%.2f' % x for x in f]}  write/particle {['%.2f' % x for x in w]}")


def last(d, key):
    """Mean of the last three dispatches (steady state of the filter) of the busiest kernel whose name has `key`."""
    best = None
    for k, v in d.items():
        if key in k and v:
            m = sum(v[-3:]) / len(v[-3:]) * KB / n
            best = m if best is None else max(best, m)
    return best


gat_f = last(fetch, "k_gather")
gat_w = last(write, "k_gather")
ff = (4.0 * W + 4.0) / gat_f if gat_f else None
wf = 4.0 * W / gat_w if gat_w else None
print("calibration: fetch factor", ff, " write factor", wf)
step_f = last(fetch, "k_step")
step_w = last(write, "k_step")
if None not in (ff, wf, step_f, step_w):
    tot = step_f * ff + step_w * wf
    print(f"k_step: fetch {step_f*ff:.2f} B + write {step_w*wf:.2f} B = {tot:.2f} B per particle")
    if out:
        json.dump({"k_step_hbm_bytes_per_particle": tot, "k_step_fetch_bytes_per_particle": step_f * ff,
                   "k_step_write_bytes_per_particle": step_w * wf, "fetch_factor": ff, "write_factor": wf,
                   "particles": n, "build": __import__("os").environ.get("EPI_BUILD", "unrecorded"),
                   "raw": {k: {kk: vv[-4:] for kk, vv in v.items()} for k, v in rep.items() if "k_step" in k or "k_gather" in k or "k_w" in k}},
                  open(out, "w"), indent=1)
