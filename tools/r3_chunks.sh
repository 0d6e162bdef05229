#!/bin/bash
python -m pytest tests/test_gpu_segment.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for ch in 1 2 4 8; do
  for kind in synth clean; do
  CK_SEG_CHUNKS=$ch python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/chunks=$ch $kind /"
  done
done
done
