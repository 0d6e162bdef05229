#!/bin/bash
# split quad fit: stage time with k_tail cut short after phase k (4: heads only, 5: selection, 6: prefix sums at the maxima, 7: pair fits + subsets, 99: all)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for s in ${STOPS:-4 5 6 7 99}; do
  CK_FIT_FLAT=1 CK_FIT_STOP_AFTER=$s python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop=$s', 'quads', d['quads'])"
done
