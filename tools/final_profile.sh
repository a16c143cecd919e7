#!/bin/bash
# The round's final measurement set on one box (GPU box, repository root): kernel statistics of the bench command, the
# default bench line, the SQ and TCC counter passes.  (gpurun merges into the local gpurun_out/: delete the local pmc_* directories of
# earlier calls before summarising with tools/pmc_valu.py / pmc_traffic.py, which read every file they find there.)  Results under gpurun_out/<round>_final (ROUND=r3 by default); copy what is kept to profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${ROUND:-r3}_final
rm -rf $O gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_fetch gpurun_out/pmc_write /tmp/prof_final
mkdir -p $O/pmc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 bench.py --no-cpu > $O/prof.log 2>&1
cp $(find /tmp/prof_final -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1
cp gpurun_out/pmc_a/*/*_counter_collection.csv $O/pmc/sq_a.csv
cp gpurun_out/pmc_b/*/*_counter_collection.csv $O/pmc/sq_b.csv
cp gpurun_out/pmc_c/*/*_counter_collection.csv $O/pmc/sq_c.csv
bash tools/pmc_traffic.sh 100000000 17 > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc_fetch/*/*_counter_collection.csv $O/pmc/tcc_fetch.csv
cp gpurun_out/pmc_write/*/*_counter_collection.csv $O/pmc/tcc_write.csv
head -8 $O/kernel_stats.csv
cat $O/bench.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'])"
