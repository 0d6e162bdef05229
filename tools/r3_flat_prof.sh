#!/bin/bash
# per-kernel durations of the detector on the bench batch with the split quad fit (CK_FIT_FLAT from the caller, default 1)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/flatprof_${1:-x}
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_detect.py 1280 800 256 3 1 > $out.log 2>&1
python3 - <<PY
import csv, glob
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Name"]
        for k in ("(anonymous namespace)::", "void "): n = n.replace(k, "")
        print(f"{n[:70]:70s} avg ms {float(r['AverageNs']) / 1e6:8.3f}  calls {r['Calls']}")
PY
