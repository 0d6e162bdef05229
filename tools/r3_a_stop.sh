#!/bin/bash
# split quad fit: time of the first kernels (k_fit<SPLIT>, all classes; k_chunk and k_tail do not run below 4) cut short after phase k
# 0: dequeue + record, 10: bounding box, 11: keys + border direction, 1: sort, 2: duplicates, 3: weights + sequence written
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for s in ${STOPS:-0 10 11 1 2 3}; do
  CK_FIT_FLAT=1 CK_FIT_STOP_AFTER=$s python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop=$s', 'quads', d['quads'])"
done
