#!/bin/bash
# per-kernel average durations of the threshold+segment stage (rocprofv3 kernel trace); usage: tools/prof_thrseg_kernels.sh <tag> [kind] [w h n]
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_${1:-x}
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_thrseg.py ${3:-1280} ${4:-800} ${5:-256} ${2:-synth} > $out.log 2>&1
python3 - <<PY
import csv, glob, re
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', r["Name"])
        if m:
            print(f"{m.group(0)[:36]:36s} avg ms {float(r['AverageNs']) / 1e6:7.3f}  calls {r['Calls']}")
PY
