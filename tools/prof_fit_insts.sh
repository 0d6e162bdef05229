#!/bin/bash
# VALU / SALU / LDS instruction counts and busy cycles of the fit kernels with every cluster stopped after phase $1 (rocprofv3 --pmc);
# prints one line per k_fit instantiation (averages per launch)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
export CK_FIT_STOP_AFTER=$1
export CK_STREAMS=1
out=$GRAFT_REPO_ROOT/gpurun_out/fitins_$1
rm -rf $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_detect.py 1280 800 256 3 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for p in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_fit" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("k_fit<")[1].split(">")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: int(kv[0].split(",")[1])):
    n = max(calls[k], 1)
    print("stop_after $1", k.replace(" ", ""), " ".join(f"{c[3:]}={v[c] / n / 1e6:.1f}M" for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")), f"waves={v['SQ_WAVES'] / n:.0f}")
PY
