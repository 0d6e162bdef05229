#!/bin/bash
# round-3 randomised parity runs on the final kernels (every script prints its own JSON summary); logs under gpurun_out/r3
mkdir -p gpurun_out/r3
( python tests/stress_segment.py 2500 361 | tail -1 | sed 's/^/segment /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
( STRESS_LOG=gpurun_out/r3/stress_detect_362.log timeout -k 10 420 python tests/stress_detect.py 500 362 | tail -1 | sed 's/^/detect /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
( STRESS_SCALE=3 STRESS_LOG=gpurun_out/r3/stress_detect_363.log timeout -k 10 400 python tests/stress_detect.py 40 363 | tail -1 | sed 's/^/detect x3 /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
( timeout -k 10 300 python tests/stress_pose.py 300 364 | tail -1 | sed 's/^/pose /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
( timeout -k 10 200 python tests/stress_batch.py 60 365 | tail -1 | sed 's/^/batch /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
( timeout -k 10 200 python tests/stress_cat.py 150 366 | tail -1 | sed 's/^/cat /' ) 2>&1 | tee -a gpurun_out/r3/stress_all.log
