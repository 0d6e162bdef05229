#!/bin/bash
# quad-fit stage time with every cluster stopped after phase k (CK_FIT_STOP_AFTER): cumulative cost of the phases, all classes
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for s in ${STOPS:-0 10 11 1 2 3 4 5 6 7 99}; do
  CK_FIT_STOP_AFTER=$s python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop=$s', 'quads', d['quads'])"
done
