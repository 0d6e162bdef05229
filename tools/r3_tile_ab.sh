#!/bin/bash
# round 3: parity of the segmentation stage, then a same-box A/B of threshold+segment against the round-2 build
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_segment.py -x -q 2>&1 | tail -5 | tee gpurun_out/r3/seg_tests.log || exit 1
python tests/stress_segment.py ${STRESS_N:-600} 301 2>&1 | tail -3 | tee gpurun_out/r3/seg_stress.log || exit 1
bash tools/ab_thrseg.sh gpurun_out/r3 chalkydri_amd/lib/ref/libchalkydri_hip_r2final.so "synth clean"
