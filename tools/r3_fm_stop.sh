#!/bin/bash
# k_fmerge by phase at 2448 x 2048 (CK_FMERGE_STOP_AFTER: 0 scan + pack, 1 + edge sweep and unions, 2 + flatten, 3 + sizes, 99 all)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
for s in ${STOPS:-0 1 2 3 99}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/fmstop_$s
  rm -rf $out
  CK_FMERGE_STOP_AFTER=$s timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_thrseg.py ${1:-2448} ${2:-2048} ${3:-128} synth > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_fmerge" in r["Name"]: print("stop=$s k_fmerge avg ms", round(float(r["AverageNs"]) / 1e6, 3))
PY
done
