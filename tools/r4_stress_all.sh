#!/bin/bash
# round-4 randomised parity runs on the final kernels (every script prints its own JSON summary) + the determinism soak; logs under gpurun_out/r4s
mkdir -p gpurun_out/r4s
( timeout -k 10 500 python tests/stress_segment.py 2500 461 | tail -1 | sed 's/^/segment /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( STRESS_LOG=gpurun_out/r4s/stress_detect_462.log timeout -k 10 420 python tests/stress_detect.py 500 462 | tail -1 | sed 's/^/detect /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( STRESS_SCALE=3 STRESS_LOG=gpurun_out/r4s/stress_detect_463.log timeout -k 10 400 python tests/stress_detect.py 60 463 | tail -1 | sed 's/^/detect x3 /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( STRESS_CAPS=1 CK_POISON=1 timeout -k 10 300 python tests/stress_detect.py 200 467 | tail -1 | sed 's/^/detect small capacities, poisoned /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( timeout -k 10 300 python tests/stress_pose.py 300 464 | tail -1 | sed 's/^/pose /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( timeout -k 10 200 python tests/stress_batch.py 60 465 | tail -1 | sed 's/^/batch /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( timeout -k 10 200 python tests/stress_cat.py 150 466 | tail -1 | sed 's/^/cat /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
( timeout -k 10 400 python tools/soak_determinism.py 300 | tail -1 | sed 's/^/soak /' ) 2>&1 | tee -a gpurun_out/r4s/stress_all.log
