#!/bin/bash
# same-box A/B of the quad fit's scheduling switches (diagnostic environment variables of k_quads.hip's launcher)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for rep in 1 2; do
  for cfg in "CK_FIT_SPLIT=1" "CK_FIT_SPLIT=3" "CK_FIT_SPLIT=2"; do
    env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$cfg', j['ms_per_step'], j.get('stage_ms_last_step'))"
  done
done
