#!/bin/bash
# cumulative cost of the k_emit phases (clusters stage time with the emit kernel cut short after phase k)
for k in 0 1 2 99; do
  CK_EMIT_STOP_AFTER=$k python tools/bench_detect.py 1280 800 256 3 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_after', $k, 'clusters ms', d['clusters'])"
done
