#!/bin/bash
# Builds a variant of the library for same-box A/B runs and profile builds: tools/build_variant.sh <name> [extra hipcc flags...]
# -> chalkydri_amd/lib/ref/libchalkydri_hip_<name>.so.  Only the translation units named in UNITS (default: k_ccl) are compiled
# with the extra flags (and -DCK_DIAG); the other objects come from the regular build (run make first).
set -e
name=$1; shift
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
src=$root/chalkydri_amd/csrc
out=$src/build/var_$name
mkdir -p $out $root/chalkydri_amd/lib/ref
units=${UNITS:-k_ccl}
objs=""
for o in $src/build/*.o; do
  b=$(basename $o .o)
  if [[ " $units " == *" $b "* ]]; then continue; fi
  if [[ -f $src/build/diag/$b.o ]]; then objs="$objs $src/build/diag/$b.o"; else objs="$objs $o"; fi
done
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I$root/include -I$src -DCK_DIAG "$@" -c $src/$u.hip -o $out/$u.o
  objs="$objs $out/$u.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/chalkydri_amd/lib/ref/libchalkydri_hip_$name.so $objs
echo built chalkydri_amd/lib/ref/libchalkydri_hip_$name.so
