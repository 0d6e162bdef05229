#!/bin/bash
# cumulative cost of the k_fit phases: run the pipeline with every cluster stopped after phase k
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for k in 0 1 2 3 4 5 6 7 99; do
  CK_FIT_STOP_AFTER=$k python tools/bench_detect.py 1280 800 256 3 1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_after', $k, 'quads_ms', d['quads'])"
done
