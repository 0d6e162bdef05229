#!/bin/bash
# k_fmerge cut short after phase N (CK_FMERGE_STOP_AFTER: 0 scan+pack, 1 unions, 2 flatten, 3 sizes, 99 all): kernel averages from rocprofv3
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
for s in ${FM_STOPS:-0 1 2 3 99}; do echo "fmerge stop=$s"; CK_FMERGE_STOP_AFTER=$s tools/prof_thrseg_kernels.sh fm$s ${1:-synth} | grep fmerge; done
