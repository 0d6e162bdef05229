#!/bin/bash
# Same-box A/B of the whole detector (stage split) over several builds of the library: tools/ab_detect.sh <lib.so>...
for rep in 1 2; do
  for lib in "$@"; do
    LIB=$lib python tools/bench_detect.py ${GEOM:-1280 800 256} 3 1 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib .so | sed s/libchalkydri_hip_//)', {k: d[k] for k in ('threshold','clusters','quads','decode','total') if k in d})"
  done
done
