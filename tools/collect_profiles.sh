#!/bin/bash
# Round profile set (run on the MI355X box through gpurun): kernel stats of the bench workload plus the two PMC passes
# behind roofline.traffic.  Usage: bash tools/collect_profiles.sh <tag>   (outputs under gpurun_out/<tag>_{stats,fetch,write})
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $root/gpurun_out/${tag}_stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $root/gpurun_out/${tag}_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $root/gpurun_out/${tag}_write.log 2>&1 &&
tail -1 $root/gpurun_out/${tag}_stats.log
