#!/bin/bash
# HBM traffic of the update's kernels from the L2 memory-side counters: FETCH_SIZE and WRITE_SIZE need separate
# passes (3 + 2 of the 4 TCC slots).  Run on the GPU box from the repository root; writes gpurun_out/pmc_fetch,
# gpurun_out/pmc_write.  Summarise with tools/pmc_traffic.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
P=${1:-100000000}
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --particles $P --steps 3 --warmup ${2:-17} --no-cpu --calibrate > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --particles $P --steps 3 --warmup ${2:-17} --no-cpu --calibrate > gpurun_out/pmc_write.log 2>&1 || exit 1
echo done
