#!/bin/bash
# cumulative cost of the k_tile phases (threshold+segment stage time with the tile kernel cut short after phase k)
for k in 0 1 2 3 4 5 6 7 8 99; do
  CK_TILE_STOP_AFTER=$k python tools/bench_thrseg.py 1280 800 256 ${1:-synth} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_after', $k, 'ms', d['ms_per_batch'])"
done
