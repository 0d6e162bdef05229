#!/bin/bash
# per-class k_fit durations with every cluster stopped after phase $1 (rocprofv3 kernel trace)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
export CK_FIT_STOP_AFTER=$1
out=$GRAFT_REPO_ROOT/gpurun_out/fitcls_$1
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_detect.py 1280 800 256 3 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_fit" in r["Name"]:
            print("stop_after $1", r["Name"][28:50], "avg ms", float(r["AverageNs"]) / 1e6)
PY
