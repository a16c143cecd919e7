#!/bin/bash
# SQ counter pass over the step kernel for several builds of the library (GPU box, repository root): tools/pmc_ab.sh base p7 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/epidemicmodeling_amd/csrc
P=${PARTICLES:-10000000}
for x in "$@"; do
  lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
  export EPI_GFX950_LIB=$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmcab_$x -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > gpurun_out/pmcab_$x.log 2>&1 || exit 1
  echo "== $x"; python3 tools/pmc_summary.py "gpurun_out/pmcab_$x/*/*_counter_collection.csv" k_step_coop
done
