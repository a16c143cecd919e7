#!/bin/bash
# VALU instruction count of each phase: SQ_INSTS_VALU of the variants of tools/phase_rep.py with the phase run once and twice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/epidemicmodeling_amd/csrc/variants
for x in ${PHASES:-A B C D}; do
  for r in 0 1; do
    rm -rf gpurun_out/prep_${x}_$r
    EPI_GFX950_LIB=$V/libepi_rep$x.so EPI_REPS=$r timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/prep_${x}_$r -- python3 bench.py --particles 10000000 --steps 2 --warmup 17 --no-cpu > gpurun_out/prep_${x}_$r.log 2>&1 || exit 1
  done
done
echo done
