#!/bin/bash
# vector / scalar / LDS instruction counts of k_tail per launch with the kernel cut short after phase k (rocprofv3 --pmc, one pass per k)
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
export CK_FIT_FLAT=1
for s in ${STOPS:-4 5 6 7 99}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/tailins_$s
  rm -rf $out
  CK_FIT_STOP_AFTER=$s timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_detect.py 1280 800 256 2 1 > /dev/null 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for p in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "${KERNEL:-k_tail}" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print("stop_after $s", {k: round(v / len(disp[k]) / 1e6, 1) for k, v in sorted(acc.items())}, "M per launch")
PY
done
