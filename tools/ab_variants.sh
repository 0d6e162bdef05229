#!/bin/bash
# times several builds of the library on one box, twice round-robin; usage: tools/ab_variants.sh <kind> lib1.so lib2.so ...
kind=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    LIB=$lib python tools/bench_thrseg.py 1280 800 256 $kind 2>/dev/null | tail -n 1 | cut -c48-72 | sed "s|^|$(basename $lib) $kind |"
  done
done
