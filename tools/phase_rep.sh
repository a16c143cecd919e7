#!/bin/bash
# Times the variants built by tools/phase_rep.py (run on the GPU box from the repository root).
V=$PWD/epidemicmodeling_amd/csrc/variants
for x in ${PHASES:-A B C D E}; do
  for r in 0 1; do
    EPI_GFX950_LIB=$V/libepi_rep$x.so EPI_REPS=$r timeout -k 10 120 python3 bench.py --steps 5 --warmup 17 --no-cpu > gpurun_out/rep_${x}_$r.json 2>/dev/null || exit 1
    echo "phase $x reps+$r $(grep -o 'kernel_ms[^,]*' gpurun_out/rep_${x}_$r.json)"
  done
done
