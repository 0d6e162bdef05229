#!/bin/bash
# CK_TILE_SWEEPS A/B on one box (0, 1, 2 pointer-jumping sweeps before the pooled unions)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for rep in 1 2; do
for sw in 1 0 2; do
  CK_TILE_SWEEPS=$sw python tools/bench_thrseg.py 1280 800 256 synth 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/sweeps=$sw /"
done
done
