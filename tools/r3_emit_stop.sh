#!/bin/bash
# clusters stage time with k_emit cut short (CK_EMIT_STOP_AFTER: 0 staging, 1 + count pass, 2 + reservation, 99 all)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for s in 0 1 2 99; do
  CK_EMIT_STOP_AFTER=$s CK_FIT_SKIP=255 python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('emit stop=$s', 'clusters', d['clusters'])"
done
