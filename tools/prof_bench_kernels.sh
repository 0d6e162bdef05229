#!/bin/bash
# per-kernel average durations of the bench workload only (the first step of tools/collect_profiles.sh); usage: tools/prof_bench_kernels.sh <tag>
tag=${1:-dev}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/${tag}_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -- python3 $root/bench.py --warmup 1 --no-cpu-baseline --no-extras --steps 5 > $root/gpurun_out/${tag}_stats.log 2>&1
f=$(find $root/gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -n 1)
cut -d, -f1-4 $f | cut -c1-150
