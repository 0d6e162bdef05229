#!/bin/bash
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for rep in 1 2; do
for st in 99 98; do
  for kind in synth clean; do
  CK_TILE_STOP_AFTER=$st python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/stop=$st $kind /"
  done
done
done
