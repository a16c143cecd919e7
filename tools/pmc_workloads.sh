#!/bin/bash
# SQ counters of the step kernel's instantiations on the other workloads next to the filter's (GPU box, repository root):
#   tools/pmc_workloads.sh <label> <program and arguments ...>
# e.g. tools/pmc_workloads.sh ens14 tools/bench_configs.py ensembles:14
# Two passes of eight SQ counters each; per kernel name: dispatches, waves and the counters per wave.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$1; shift
rm -rf gpurun_out/pmcw_a gpurun_out/pmcw_b
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmcw_a -- python3 "$@" > gpurun_out/pmcw_a.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d gpurun_out/pmcw_b -- python3 "$@" > gpurun_out/pmcw_b.log 2>&1 || exit 1
python3 - "$L" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcw_[ab]/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_step" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    # the second half of the dispatches (the first run of a case warms up)
    half = {n: v[len(v) // 2:] for n, v in c.items()}
    waves = sum(half["SQ_WAVES"]) / len(half["SQ_WAVES"])
    line = " ".join(f"{n[3:]}={sum(v) / len(v) / waves:.0f}" for n, v in sorted(half.items()) if n != "SQ_WAVES")
    print(f"{sys.argv[1]} | {k} | dispatches {len(c['SQ_WAVES'])} waves {waves:.0f} | {line}")
PY
