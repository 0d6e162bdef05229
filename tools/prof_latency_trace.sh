#!/bin/bash
# kernel timeline of one frame per call (rocprofv3 kernel trace over tools/bench_latency_stages.py); args: noise decimate
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/lat_trace
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_latency_stages.py ${1:-3} ${2:-2} > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_tile" in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]          # the last complete frame
t0 = int(rows[s]["Start_Timestamp"]); prev = t0; busy = 0
for r in rows[s - 3:e - 3]:
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?|__amd_rocclr_\w+)", r["Kernel_Name"]); nm = (m.group(0) if m else r["Kernel_Name"])[:34]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += en - st
    print(f"{nm:34s} start {(st - t0) / 1e3:8.1f} us  dur {(en - st) / 1e3:7.1f}  gap {(st - prev) / 1e3:6.1f}")
    prev = en
print("busy us", busy / 1e3)
PY
