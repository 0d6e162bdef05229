#!/bin/bash
# threshold+segment on the bench-sized batch with k_fmerge's per-workgroup root capacity lowered (CK_FMERGE_CAP): a frame whose roots
# no longer fit one workgroup is joined by two, one per colour, each with half the LDS
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for cap in ${CAPS:-0 16384 14336 12288 11264}; do
  for kind in synth clean; do
    if [ $cap = 0 ]; then python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c1-200 | sed "s/^/cap=default $kind /"
    else CK_FMERGE_CAP=$cap python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c1-200 | sed "s/^/cap=$cap $kind /"; fi
  done
done
