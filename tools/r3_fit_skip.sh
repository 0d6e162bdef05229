#!/bin/bash
# quad-fit stage time with some size classes not launched (CK_FIT_SKIP mask): what each class costs beside the others
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for m in ${MASKS:-0 128 1 64 2 4 8 16}; do
  CK_FIT_SKIP=$m python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip=$m', 'quads', d['quads'], 'clusters', d['clusters'], 'thr', d['threshold'], 'total', d.get('total'))"
done
