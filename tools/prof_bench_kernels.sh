#!/bin/bash
# per-kernel average durations of the bench workload (rocprofv3 kernel trace); usage: bash tools/prof_bench_kernels.sh <tag>
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_${1:-x}
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS > /dev/null 2>&1
python3 - <<PY
import csv, glob, re
for p in glob.glob("$out/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', r["Name"])
        if m and float(r["Percentage"]) > 0.4:
            print(f"{m.group(0)[:36]:36s} avg ms {float(r['AverageNs']) / 1e6:7.3f}  calls {r['Calls']}")
PY
