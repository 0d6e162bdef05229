#!/bin/bash
# split quad fit (CK_FIT_FLAT): parity of the forced split path on the detector tests, then the stage time with and without it
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
set -o pipefail
export CK_FIT_FLAT=2
timeout -k 10 600 python -m pytest tests/test_gpu_detect.py -x -q -m gpu -k "quads or detect_matches or adversarial or small_and_ragged or golden" > gpurun_out/flat_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/flat_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tests/stress_detect.py 150 77 > gpurun_out/flat_stress.log 2>&1; rc=$?; tail -n 3 gpurun_out/flat_stress.log
[ $rc -ne 0 ] && exit $rc
for f in 0 1; do
  CK_FIT_FLAT=$f timeout -k 10 200 python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -n 1 > gpurun_out/flat_bench_$f.json && cat gpurun_out/flat_bench_$f.json
done
