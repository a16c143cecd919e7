#!/bin/bash
# Memory-pipeline and front-end counter passes over the step kernel (GPU box, repository root): which unit besides the
# vector ALU is close to its limit.  tools/pmc_bound.sh [variant]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/epidemicmodeling_amd/csrc
x=${1:-base}
lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
export EPI_GFX950_LIB=$lib
P=${PARTICLES:-10000000}
i=0
for set in \
  "GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" \
  "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcb_${x}_$i -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmcb_${x}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmcb_${x}_$i.log; exit 1; }
  python3 tools/pmc_summary.py "gpurun_out/pmcb_${x}_$i/*/*_counter_collection.csv" k_step_coop
done
