#!/bin/bash
# like tools/ab.sh, but prints the time of an update outside its step kernel too
V=$PWD/epidemicmodeling_amd/csrc
for x in "$@"; do
  lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
  EPI_GFX950_LIB=$lib timeout -k 10 120 python3 bench.py --steps ${STEPS:-5} --warmup ${WARMUP:-17} --particles ${PARTICLES:-100000000} --no-cpu > gpurun_out/ab_$x.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/ab_$x.json')); print('$x', 'step kernel %.3f ms' % d['roofline']['kernel_ms'], 'rest of the update %.3f ms' % (d['ms_per_step'] - d['roofline']['kernel_ms']))"
done
