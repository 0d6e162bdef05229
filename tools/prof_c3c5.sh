#!/bin/bash
# kernel durations of the two larger BASELINE configs only (C3 1920x1080x512, C5 2448x2048x256): tools/prof_c3c5.sh <tag> [batch3 batch5]
tag=${1:-dev}; b3=${2:-512}; b5=${3:-256}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $root/gpurun_out/${tag}_c3 $root/gpurun_out/${tag}_c5
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c3 -- $B --steps 2 --width 1920 --height 1080 --batch $b3 --tags 30 --unique 64 > $root/gpurun_out/${tag}_c3.log 2>&1 && echo "c3 done" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c5 -- $B --steps 2 --width 2448 --height 2048 --batch $b5 --tags 20 --unique 32 > $root/gpurun_out/${tag}_c5.log 2>&1 && echo "c5 done"
for t in c3 c5; do f=$(find $root/gpurun_out/${tag}_$t -name '*kernel_stats.csv' | head -n 1); echo "== $t"; grep -E "k_tile|k_fmerge|k_fseam|k_fapply" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://' | cut -c1-40,200-260; done
