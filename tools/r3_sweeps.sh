#!/bin/bash
# CK_TILE_SWEEPS A/B on one box (0, 1, 2 pointer-jumping sweeps before the pooled unions)
for rep in 1 2; do
for sw in 1 0 2; do
  CK_TILE_SWEEPS=$sw python tools/bench_thrseg.py 1280 800 256 synth 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/sweeps=$sw /"
done
done
