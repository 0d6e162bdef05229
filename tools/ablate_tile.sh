#!/bin/bash
# Where k_tile's time goes: threshold+segment timed with the kernel cut short after phase N (CK_TILE_STOP_AFTER), dense-noise and
# low-noise batches.  usage: tools/ablate_tile.sh <outdir> [stops...]
out=${1:-gpurun_out/ablate}; shift
stops=${@:-0 1 2 3 4 5 6 7 99}
mkdir -p "$out"
for kind in synth clean; do
  for s in $stops; do
    CK_TILE_STOP_AFTER=$s python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | sed "s/^/stop=$s /" >> "$out/$kind.log"
  done
done
cat "$out"/synth.log "$out"/clean.log
