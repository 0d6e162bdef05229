#!/bin/bash
# Same-box A/B of the threshold+segment stage: the reference build (chalkydri_amd/lib/ref/*.so) against the working tree,
# alternating (ref, new, ref, new) so that clock drift of the box shows up as a difference between the repeats.
# usage: tools/ab_thrseg.sh <outdir> [ref.so] [kinds]
out=${1:-gpurun_out/ab}; ref=${2:-chalkydri_amd/lib/ref/libchalkydri_hip_r1tile.so}; kinds=${3:-synth clean}
mkdir -p "$out"
for kind in $kinds; do
  for rep in 1 2; do
    LIB=$ref python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/ref $kind /"
    python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-85 | sed "s/^/new $kind /"
  done
done | tee "$out/ab.log"
