#!/bin/bash
# Kernel statistics of the two sharded forms on one GPU (GPU box, repository root): the in-library update with 8 shards
# sharing the device, and the one-process-per-GPU update with a process group of one.  Results under gpurun_out/r2_sharded.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r2_sharded
rm -rf $O /tmp/prof_node /tmp/prof_sh && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_node -- python3 tools/bench_node_filter.py 100000000 8 10 > $O/node8.log 2>&1
cp $(find /tmp/prof_node -name '*kernel_stats.csv' | head -1) $O/node8_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh -- python3 bench.py --sharded --no-cpu --steps 10 > $O/sharded1.log 2>&1
cp $(find /tmp/prof_sh -name '*kernel_stats.csv' | head -1) $O/sharded1_kernel_stats.csv
grep "node filter" $O/node8.log; grep -o '"ms_per_step": [^,]*' $O/sharded1.log
head -14 $O/node8_kernel_stats.csv | cut -c1-150; head -12 $O/sharded1_kernel_stats.csv | cut -c1-150
