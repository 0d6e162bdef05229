#!/bin/bash
# same-box A/B of two builds of the library on the bench workload; usage: tools/ab_libs.sh ref.so [new.so] [bench.py flags]
ref=$(realpath $1); new=$(realpath ${2:-chalkydri_amd/lib/libchalkydri_hip.so}); shift; shift
for rep in 1 2 3; do
  for lib in $ref $new; do
    CHALKYDRI_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -n 1 |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['stage_ms_last_step']; print('$(basename $lib)', j['ms_per_step'], 'clusters', s['clusters'], 'quads', s['quads'])"
  done
done
