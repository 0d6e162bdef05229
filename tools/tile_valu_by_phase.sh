#!/bin/bash
# VALU / SALU / LDS instruction counts of k_tile cut short after each phase (CK_TILE_STOP_AFTER): differences = per-phase counts.
# usage (through gpurun): [PMC="SQ_..."] tools/tile_valu_by_phase.sh [kind]
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
kind=${1:-synth}
pmc=${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY}
for s in ${STOPS:-0 1 2 4 5 6 65 7 99}; do
  out=$root/gpurun_out/valu_$s
  rm -rf $out
  CK_TILE_STOP_AFTER=$s timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $out -- python3 $root/tools/bench_thrseg.py 1280 800 256 $kind > /dev/null 2>&1
  python3 - <<PY
import csv, glob
acc, n = {}, {}
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_tile" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
w = 64000 * 4
print("stop=$s", {k: round(v / len(n[k]) / w, 1) for k, v in sorted(acc.items())})
PY
done
