#!/bin/bash
# tools/wide_state.py with each of the named library builds (names as for tools/ab.sh)
V=$PWD/epidemicmodeling_amd/csrc
for x in "$@"; do
  lib=$V/variants/libepi_$x.so; [ "$x" = base ] && lib=$V/libepi_gfx950.so
  echo "== $x"; EPI_GFX950_LIB=$lib timeout -k 10 200 python3 tools/wide_state.py || exit 1
done
