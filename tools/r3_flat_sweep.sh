#!/bin/bash
# quads stage time of the bench batch: unsplit fit, split fit, and knobs of the split fit
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
run() { env "$@" python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'quads', d['quads'], 'total', d['total'])"; }
for k in "$@"; do run CK_FIT_FLAT=1 $k; done
