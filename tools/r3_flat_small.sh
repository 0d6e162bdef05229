#!/bin/bash
# unsplit against split quad fit on smaller quad-stage workloads (where does the split start to pay?)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
run() { env "$1" python tools/bench_detect.py ${@:2} 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'quads', d['quads'], 'total', d['total'])"; }
for f in 0 2; do
  run CK_FIT_FLAT=$f 1280 800 256 3 2
  run CK_FIT_FLAT=$f 1280 800 64 3 1
  run CK_FIT_FLAT=$f 1280 800 32 3 1
  run CK_FIT_FLAT=$f 640 480 256 3 1
done
