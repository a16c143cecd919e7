#!/bin/bash
# Times variant builds of the library against each other (GPU box, repository root):  tools/ab.sh w2 n3 ...
# "base" is the in-tree libepi_gfx950.so; other names are csrc/variants/libepi_<name>.so (make variant NAME=...).
V=$PWD/epidemicmodeling_amd/csrc
for x in "$@"; do
  lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
  EPI_GFX950_LIB=$lib timeout -k 10 120 python3 bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-17} --particles ${PARTICLES:-100000000} --no-cpu > gpurun_out/ab_$x.json 2>/dev/null || exit 1
  echo "$x $(grep -o 'kernel_ms[^,]*' gpurun_out/ab_$x.json) $(grep -o '"value": [^,]*' gpurun_out/ab_$x.json)"
done
