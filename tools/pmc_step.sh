#!/bin/bash
# Collects SQ counters of the fused step kernel on a 10^7-particle bench run (two passes, 8 SQ counters each).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-10000000}
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_a -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d gpurun_out/pmc_b -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_b.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d gpurun_out/pmc_c -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmc_c.log 2>&1 || exit 1
echo done
