#!/bin/bash
# SQ counters of the fused step kernel for a variant build (GPU box, repository root):  tools/pmc_variant.sh <name|base> [particles]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/epidemicmodeling_amd/csrc
lib=$V/variants/libepi_$1.so; [ "$1" = base ] && lib=$V/libepi_gfx950.so
export EPI_GFX950_LIB=$lib
P=${2:-10000000}
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_a -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d gpurun_out/pmc_b -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_b.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_[ab]/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "k_step_coop<true" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean={k:sum(v[-2:])/len(v[-2:]) for k,v in acc.items()}
w=mean["SQ_WAVES"]
print(" ".join(f"{k[3:]}={v/w:.0f}" for k,v in sorted(mean.items()) if k!="SQ_WAVES"))
PY
