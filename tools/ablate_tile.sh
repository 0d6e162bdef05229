#!/bin/bash
# Where k_tile's time goes: threshold+segment timed with the kernel cut short after phase N (CK_TILE_STOP_AFTER), dense-noise and
# low-noise batches.  usage: tools/ablate_tile.sh <outdir> [stops...]
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
out=${1:-gpurun_out/ablate}; shift
stops=${@:-0 1 2 3 4 5 6 7 99}
mkdir -p "$out"
for kind in synth clean; do
  for s in $stops; do
    CK_TILE_STOP_AFTER=$s python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | sed "s/^/stop=$s /" >> "$out/$kind.log"
  done
done
cat "$out"/synth.log "$out"/clean.log
