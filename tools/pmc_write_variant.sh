#!/bin/bash
# HBM bytes written per particle by the fused step kernel for variant builds (GPU box, repository root):
#   tools/pmc_write_variant.sh base r4 r5      ("base" = the in-tree library, others csrc/variants/libepi_<name>.so)
# The kernel's algorithmic writes are 72 B per particle; what a build writes beyond that is stack traffic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/epidemicmodeling_amd/csrc
P=${PARTICLES:-10000000}
for x in "$@"; do
  lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
  export EPI_GFX950_LIB=$lib
  rm -rf gpurun_out/pmc_wv
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wv -- python3 bench.py --particles $P --steps 3 --warmup 5 --no-cpu > gpurun_out/pmc_wv.log 2>&1 || exit 1
  python3 - "$x" "$P" <<'PY'
import csv, glob, sys
v = []
for f in glob.glob("gpurun_out/pmc_wv/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_step_coop<true" in row["Kernel_Name"] and row["Counter_Name"] == "WRITE_SIZE":
            v.append(float(row["Counter_Value"]) * 1024.0 / float(sys.argv[2]))
print(sys.argv[1], "write B/particle (last three launches):", " ".join("%.1f" % x for x in v[-3:]), "scratch", row["Scratch_Size"] if v else "?")
PY
done
