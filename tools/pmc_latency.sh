#!/bin/bash
# Average latencies (LDS, scalar memory, vector memory, instruction fetch) and per-class busy cycles of the step kernel
# (GPU box, repository root): tools/pmc_latency.sh [particles].  One counter pass per derived metric.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=${1:-10000000}
O=gpurun_out/pmc_lat
rm -rf $O; mkdir -p $O
for m in LdsLatency SmemLatency VmemLatency InstrFetchLatency; do
  timeout -s KILL 120 rocprofv3 --pmc $m --output-format csv -d $O/$m -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > $O/$m.log 2>&1 || { echo "$m failed"; tail -3 $O/$m.log; }
done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM --output-format csv -d $O/busy -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > $O/busy.log 2>&1 || echo "busy failed"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_WAVE_CYCLES --output-format csv -d $O/lds -- python3 bench.py --particles $P --steps 2 --warmup 1 --no-cpu > $O/lds.log 2>&1 || echo "lds failed"
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_lat/*/*/*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "k_step_coop<true" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
mean={k:sum(v[-2:])/len(v[-2:]) for k,v in acc.items()}
w=mean.get("SQ_WAVES",1)
for k,v in sorted(mean.items()):
    print(k, f"{v:.1f}" if "Latency" in k else f"{v/w:.1f} per wave")
PY
