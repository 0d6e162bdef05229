for k in 0 10 11 1; do
  CK_FIT_STOP_AFTER=$k python tools/bench_detect.py 1280 800 256 3 1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_after', $k, 'quads_ms', d['quads'])"
done
