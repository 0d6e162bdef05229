#!/bin/bash
python -m pytest tests/test_gpu_segment.py -x -q 2>&1 | tail -2
python tests/stress_segment.py 400 331 2>&1 | tail -1
for rep in 1 2; do
  LIB=chalkydri_amd/lib/ref/libchalkydri_hip_r2final.so python tools/bench_thrseg.py 2448 2048 128 synth 2>/dev/null | tail -n 1 | cut -c48-100 | sed "s/^/ref c5 /"
  python tools/bench_thrseg.py 2448 2048 128 synth 2>/dev/null | tail -n 1 | cut -c48-100 | sed "s/^/new c5 /"
  LIB=chalkydri_amd/lib/ref/libchalkydri_hip_r2final.so python tools/bench_thrseg.py 1920 1080 256 synth 2>/dev/null | tail -n 1 | cut -c48-100 | sed "s/^/ref c3 /"
  python tools/bench_thrseg.py 1920 1080 256 synth 2>/dev/null | tail -n 1 | cut -c48-100 | sed "s/^/new c3 /"
done
