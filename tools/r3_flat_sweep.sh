#!/bin/bash
# quads stage time of the bench batch: unsplit fit, split fit, and the split fit's grid knobs
run() { env "$@" python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'quads', d['quads'], 'total', d['total'])"; }
run CK_FIT_FLAT=0
run CK_FIT_FLAT=1
run CK_FIT_FLAT=1 CK_TAIL_WGS=8
run CK_FIT_FLAT=1 CK_TAIL_WGS=32
run CK_FIT_FLAT=1 CK_CHUNK_WGS=3
run CK_FIT_FLAT=1 CK_CHUNK_WGS=6
run CK_FIT_FLAT=1 CK_CHUNK_WGS=64
run CK_FIT_FLAT=1 CK_FIT_TAILS_ASIDE=0
