#!/bin/bash
for rep in 1 2; do
for st in 99 98; do
  for kind in synth clean; do
  CK_TILE_STOP_AFTER=$st python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/stop=$st $kind /"
  done
done
done
