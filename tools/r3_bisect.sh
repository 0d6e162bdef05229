#!/bin/bash
# the 2448x2048 detector tests under the split fit with k_tail built for four / five waves per SIMD, and the stage time of both
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for k in "X=0" "CK_TAIL_WPS=5"; do
  for rep in 1 2; do
    env CK_FIT_FLAT=2 $k timeout -k 10 300 python -m pytest tests/test_gpu_detect.py -x -q -m gpu -k "2448 or 1920" 2>&1 | tail -n 1 | sed "s/^/$k rep $rep: /"
  done
done
bash tools/r3_flat_sweep.sh X=0 CK_TAIL_WPS=5
