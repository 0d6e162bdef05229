#!/bin/bash
# clusters stage time of the bench batch under the given environment settings (one per argument)
run() { env "$@" python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', 'clusters', d['clusters'], 'total', d['total'])"; }
for k in "$@"; do run $k; done
