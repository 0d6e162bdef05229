#!/bin/bash
# unsplit against split quad fit on the other configurations: 1920x1080 x 256 (30 tags), 2448x2048 x 128 (20 tags), quad_decimate 2
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
run() { env "$1" python tools/bench_detect.py ${@:2} 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'quads', d['quads'], 'total', d['total'], 'dets', d['dets_per_frame'])"; }
for f in 0 1; do
  run CK_FIT_FLAT=$f 1920 1080 256 3 1
  run CK_FIT_FLAT=$f 2448 2048 128 3 1
  run CK_FIT_FLAT=$f 1280 800 256 3 2
done
