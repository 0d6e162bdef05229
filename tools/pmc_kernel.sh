#!/bin/bash
# per-launch averages of SQ/TCC counters for the kernels whose name contains <substr>, on the threshold+segment bench workload.
# usage (through gpurun): tools/pmc_kernel.sh <substr> <kind> COUNTER...   (at most 8 SQ counters per pass)
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
sub=$1; kind=$2; shift 2
out=$root/gpurun_out/pmc_k
rm -rf $out
timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $root/tools/bench_thrseg.py ${GEOM:-1280 800 256} $kind > /dev/null 2>&1
python3 - <<PY
import csv, glob
acc, n = {}, {}
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$sub" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
print("$sub $kind per launch:", {k: round(v / len(n[k])) for k, v in sorted(acc.items())})
PY
