#!/bin/bash
# randomised parity stress of the split quad fit (forced on every call: CK_FIT_FLAT=2): plain, large frames (every size class, clusters
# over several spans), undersized + poisoned buffers, pose, batch composition.  Writes gpurun_out/flat_stress_*.log
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
export CK_FIT_FLAT=2
( timeout -k 10 500 python tests/stress_detect.py ${N1:-800} 401 > gpurun_out/flat_stress_plain.log 2>&1; echo "plain rc=$?" ) &
( STRESS_SCALE=3 timeout -k 10 500 python tests/stress_detect.py ${N2:-120} 402 > gpurun_out/flat_stress_large.log 2>&1; echo "large rc=$?" ) &
( STRESS_CAPS=1 CK_POISON=1 timeout -k 10 500 python tests/stress_detect.py ${N3:-500} 403 > gpurun_out/flat_stress_caps.log 2>&1; echo "caps rc=$?" ) &
( timeout -k 10 500 python tests/stress_pose.py ${N4:-150} 404 > gpurun_out/flat_stress_pose.log 2>&1; echo "pose rc=$?" ) &
wait
tail -n 2 gpurun_out/flat_stress_plain.log gpurun_out/flat_stress_large.log gpurun_out/flat_stress_caps.log gpurun_out/flat_stress_pose.log
