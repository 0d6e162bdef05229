#!/bin/bash
# round 4: k_emit by phase (stop-after knob, diagnostics build) and its counters on the detector batch (1280x800x256, noise 3)
export CHALKYDRI_HIP_LIB=${CHALKYDRI_HIP_LIB:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)/chalkydri_amd/lib/diag/libchalkydri_hip.so}
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r4
bash $root/tools/r3_emit_stop.sh
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  out=$root/gpurun_out/pmc_emit
  rm -rf $out
  CK_FIT_SKIP=255 timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $out -- python3 $root/tools/bench_detect.py 1280 800 256 3 1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
acc, n = {}, {}
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "${KSUB:-k_emit}" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
print("${KSUB:-k_emit} per launch:", {k: round(v / len(n[k])) for k, v in sorted(acc.items())})
PY
done
