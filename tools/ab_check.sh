#!/bin/bash
# parity of the segmentation stage on a variant build (CHALKYDRI_HIP_LIB), then nothing else: tools/ab_check.sh <lib.so> [stress cases]
set -o pipefail
export CHALKYDRI_HIP_LIB=$(realpath $1)
python -m pytest tests/test_gpu_segment.py -x -q 2>&1 | tail -2 || exit 1
python tests/stress_segment.py ${2:-300} 401 2>&1 | tail -2 || exit 1
