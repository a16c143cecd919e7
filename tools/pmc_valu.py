#!/usr/bin/env python3
"""Instruction mix of the fused step kernel from the SQ counter passes of tools/pmc_step.sh:
mean per wavefront of the last two dispatches, written as JSON (bench.py reports it next to the HBM roofline).

    python3 tools/pmc_valu.py profiles/pmc_valu.json
"""
import collections, csv, glob, json, sys

acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_[abc]/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "k_step_coop<true" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean = {k: sum(v[-2:]) / len(v[-2:]) for k, v in acc.items()}
waves = mean["SQ_WAVES"]
per_wave = {k[len("SQ_INSTS_"):].lower(): round(v / waves, 1) for k, v in mean.items() if k.startswith("SQ_INSTS_")}
import os
out = {"kernel": "k_step_coop<true, false, 14, true, true, 1>", "build": os.environ.get("EPI_BUILD", "unrecorded"),
       "waves_per_dispatch": waves, "insts_per_wave_step": per_wave, "valu_insts_per_wave_step": per_wave["valu"],
       "other_counters_per_wave_step": {k[3:].lower(): round(v / waves, 1) for k, v in mean.items() if not k.startswith("SQ_INSTS_") and k != "SQ_WAVES"}}
json.dump(out, open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout, indent=1)
print(json.dumps(out))
