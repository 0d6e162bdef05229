#!/bin/bash
# guard-page runs (CK_POISON=3: every device buffer ends at an unmapped granule): the reproducer of the runtime's address reuse,
# the fp64 probe test, and the random-capacity detector stress.  usage: tools/r3_guard_stress.sh <cases> <seed>
mkdir -p gpurun_out/r3
timeout -k 5 60 tools/probes/vmm_reuse_probe 6 1 0 | tail -2
timeout -k 5 60 tools/probes/vmm_reuse_probe 6 1 1 | tail -2
CK_POISON=3 python -m pytest tests/test_gpu_segment.py -x -q -k fp64 2>&1 | tail -2
CK_POISON=3 STRESS_CAPS=1 STRESS_LOG=gpurun_out/r3/guard_stress_$2.log timeout -k 10 1000 python tests/stress_detect.py $1 $2 2>&1 | tail -3
