#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_*/*/*_counter_collection.csv"
want = sys.argv[2] if len(sys.argv) > 2 else "k_step"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(pat)):
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        # the first dispatches are warm-up; report the mean of the last two
        tail = v[-2:]
        print(f"   {c:32s} {sum(tail)/len(tail):16.0f}   (n={len(v)})")
