#!/bin/bash
# Round profile set (run on the MI355X box through gpurun): kernel stats of the bench workload plus the two PMC passes
# behind roofline.traffic, and kernel stats of the two larger BASELINE configs (C3, C5).
# Usage: tools/collect_profiles.sh <tag>   (outputs under gpurun_out/<tag>_{stats,fetch,write,c3,c5})
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --warmup 1 --no-cpu-baseline --no-extras"
rm -rf $root/gpurun_out/${tag}_stats $root/gpurun_out/${tag}_fetch $root/gpurun_out/${tag}_write $root/gpurun_out/${tag}_c3 $root/gpurun_out/${tag}_c5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -- $B --steps 5 > $root/gpurun_out/${tag}_stats.log 2>&1 && echo "stats done" &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_fetch -- $B --steps 2 > $root/gpurun_out/${tag}_fetch.log 2>&1 && echo "fetch done" &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_write -- $B --steps 2 > $root/gpurun_out/${tag}_write.log 2>&1 && echo "write done" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c3 -- $B --steps 2 --width 1920 --height 1080 --batch 512 --tags 30 --unique 64 > $root/gpurun_out/${tag}_c3.log 2>&1 && echo "c3 done" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_c5 -- $B --steps 2 --width 2448 --height 2048 --batch 256 --tags 20 --unique 32 > $root/gpurun_out/${tag}_c5.log 2>&1 && echo "c5 done"
tail -n 1 $root/gpurun_out/${tag}_stats.log $root/gpurun_out/${tag}_c3.log $root/gpurun_out/${tag}_c5.log
