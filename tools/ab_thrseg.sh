#!/bin/bash
# Same-box A/B of the threshold+segment stage: the reference build (chalkydri_amd/lib/ref/*.so) against the working tree.
# usage: tools/ab_thrseg.sh <outdir> [ref.so]
out=${1:-gpurun_out/ab}; ref=${2:-chalkydri_amd/lib/ref/libchalkydri_hip_r1tile.so}
mkdir -p "$out"
for kind in synth clean; do
  LIB=$ref python tools/bench_thrseg.py 1280 800 256 $kind > "$out/ref_$kind.log" 2>&1
  python tools/bench_thrseg.py 1280 800 256 $kind > "$out/new_$kind.log" 2>&1
done
tail -n 1 "$out"/ref_*.log "$out"/new_*.log
