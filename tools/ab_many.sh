#!/bin/bash
# Same-box A/B of threshold+segment over several builds of the library, alternating (a, b, c, a, b, c) so that drift of the box
# shows as a difference between the repeats.  usage: tools/ab_many.sh <kind> <lib.so>...   (kind: synth | clean | bench | bench_quiet)
kind=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    LIB=$lib python tools/bench_thrseg.py ${GEOM:-1280 800 256} $kind 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib .so | sed s/libchalkydri_hip_//)', '$kind', d['ms_per_batch'], d['frac_of_8TBps'])"
  done
done
