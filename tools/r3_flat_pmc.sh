#!/bin/bash
# SQ counters of the split quad fit's kernels on the bench batch (two rocprofv3 --pmc passes over tools/bench_detect.py); per-launch averages
# (the knobs this script sets exist only in the diagnostics build of the library: ck_internal.h, CK_KNOB)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
export CK_FIT_FLAT=1
for pass in 1 2; do
  out=$root/gpurun_out/flatpmc_$pass
  rm -rf $out
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES"
  else C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; fi
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $out -- python3 $root/tools/bench_detect.py 1280 800 256 2 1 > /dev/null 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$root/gpurun_out/flatpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("k_tail", "k_chunk", "k_seq", "k_emit", "k_scatter")): continue
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(n, r["Counter_Name"])].add(r["Dispatch_Id"])
for n, c in sorted(acc.items()):
    print(n, {k: round(v / len(disp[(n, k)]) / 1e6, 2) for k, v in sorted(c.items())}, "(millions per launch)")
PY
