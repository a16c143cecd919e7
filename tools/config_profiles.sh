#!/bin/bash
# Wall times and kernel statistics of the other configurations (closed loop, scoring, ensembles) on one box (GPU box, repository
# root); results under gpurun_out/r3_cfg/c2, c4, c5: copy what is kept to profiles/r3_config2, 4, 5.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_cfg; rm -rf $O; mkdir -p $O/c2 $O/c4 $O/c5
timeout -k 10 200 python3 tools/bench_closed_loop.py > $O/c2/bench_closed_loop.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc2 -- python3 tools/bench_configs.py closed_loop > $O/c2/prof.log 2>&1
cp $(find /tmp/pc2 -name '*kernel_stats.csv' | head -1) $O/c2/kernel_stats.csv
timeout -k 10 200 python3 tools/bench_configs.py scoring > $O/c4/bench_configs.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc4 -- python3 tools/bench_configs.py scoring > $O/c4/prof.log 2>&1
cp $(find /tmp/pc4 -name '*kernel_stats.csv' | head -1) $O/c4/kernel_stats.csv
timeout -k 10 300 python3 tools/bench_configs.py ensembles > $O/c5/bench_configs.log 2>&1
timeout -k 10 300 python3 tools/wide_state.py > $O/c5/wide_state.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -- python3 tools/bench_configs.py ensembles > $O/c5/prof.log 2>&1
cp $(find /tmp/pc5 -name '*kernel_stats.csv' | head -1) $O/c5/kernel_stats.csv
cat $O/c2/bench_closed_loop.log $O/c4/bench_configs.log $O/c5/bench_configs.log
