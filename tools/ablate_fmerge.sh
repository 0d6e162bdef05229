#!/bin/bash
# k_fmerge cut short after phase N (CK_FMERGE_STOP_AFTER: 0 scan+pack, 1 unions, 2 flatten, 3 sizes, 99 all): kernel averages from rocprofv3
for s in ${FM_STOPS:-0 1 2 3 99}; do echo "fmerge stop=$s"; CK_FMERGE_STOP_AFTER=$s tools/prof_thrseg_kernels.sh fm$s ${1:-synth} | grep fmerge; done
