#!/bin/bash
# round 4: counters of EVERY kernel of the detector batch (1280x800x256, noise 3), per launch and per wave, two passes:
# tools/pmc_all.sh [ENV=...]  -> gpurun_out/r4/pmc_all.txt.  (The program itself stands after `--`, never a shell.)
root=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
mkdir -p $root/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA"; do
  out=$root/gpurun_out/pmc_all_$i; i=$((i+1))
  rm -rf $out
  timeout -k 10 250 rocprofv3 --pmc $pmc --output-format csv -d $out -- python3 $root/tools/bench_detect.py ${GEOM:-1280 800 256} 3 1 > /dev/null 2>&1
done
python3 - <<PY | tee $root/gpurun_out/r4/pmc_all${TAG:+_$TAG}.txt
import csv, glob, re
acc, n = {}, {}
for f in glob.glob("$root/gpurun_out/pmc_all_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:44]
        g = int(r.get("Grid_Size", 0) or 0)
        key = (k, g)
        acc.setdefault(key, {}); n.setdefault(key, {})
        acc[key][r["Counter_Name"]] = acc[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[key].setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
print("per LAUNCH averages; wave figures = counter / SQ_WAVES; cycle counters are in units of 4 clocks summed over waves (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_*) ")
for key in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    a = {c: v / len(n[key][c]) for c, v in acc[key].items()}
    w = a.get("SQ_WAVES", 0) or 1
    if a.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    print(f"{key[0]:44s} grid {key[1]:>9d} waves {w:9.0f} | per wave: VALU {a.get('SQ_INSTS_VALU',0)/w:8.0f} SALU {a.get('SQ_INSTS_SALU',0)/w:7.0f} LDS {a.get('SQ_INSTS_LDS',0)/w:7.0f} VMEMrd {a.get('SQ_INSTS_VMEM_RD',0)/w:6.0f} wr {a.get('SQ_INSTS_VMEM_WR',0)/w:6.0f} | wave-cyc(x4) {a.get('SQ_WAVE_CYCLES',0)/w:9.0f} wait {a.get('SQ_WAIT_ANY',0)/max(a.get('SQ_WAVE_CYCLES',1),1):5.2f} | activeVALU(x4) {a.get('SQ_ACTIVE_INST_VALU',0)/w:8.0f} activeLDS {a.get('SQ_ACTIVE_INST_LDS',0)/w:7.0f} ldsIdx {a.get('SQ_LDS_IDX_ACTIVE',0)/w:7.0f} conflict {a.get('SQ_LDS_BANK_CONFLICT',0)/w:7.0f} busy {a.get('SQ_BUSY_CYCLES',0):10.0f}")
PY
