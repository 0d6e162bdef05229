#!/bin/bash
# cumulative cost of the k_emit phases (clusters stage time with the emit kernel cut short after phase k)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for k in 0 1 2 99; do
  CK_EMIT_STOP_AFTER=$k python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_after', $k, 'clusters ms', d['clusters'])"
done
