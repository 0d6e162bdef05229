#!/bin/bash
# the large fit classes with their keys in global memory (CK_FIT_GK mask) against the LDS-resident variants, same box
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for rep in 1 2; do
for m in 0 8 16 24; do
  CK_FIT_GK=$m python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gk=$m', 'quads', d['quads'], 'dets', d['dets_per_frame'])"
done
done
