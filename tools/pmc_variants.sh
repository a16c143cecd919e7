#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_v
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/pmc_v -- python3 tools/phase_cost.py 1 > gpurun_out/pmc_v.log 2>&1 || exit 1
echo done
