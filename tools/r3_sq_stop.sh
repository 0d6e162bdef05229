#!/bin/bash
# k_sqpnp by phase (CK_SQ_STOP_AFTER: 1 after Omega, 2 after the eigen-decomposition, 3 after the refinements, 99 all): kernel durations
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 99; do
  out=$GRAFT_REPO_ROOT/gpurun_out/sqstop_$s
  rm -rf $out
  CK_SQ_STOP_AFTER=$s timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_sqpnp" in r["Name"] or "k_glue" in r["Name"] or "k_measure" in r["Name"]: print("sq stop=$s", r["Name"][:40], "avg ms", round(float(r["AverageNs"]) / 1e6, 4))
PY
done
