#!/bin/bash
# VALU / SALU instruction counts of the fit kernels with every cluster stopped after phase $1 (rocprofv3 --pmc)
cd /tmp && export TMPDIR=/tmp
export CK_FIT_STOP_AFTER=$1
export CK_STREAMS=1
out=$GRAFT_REPO_ROOT/gpurun_out/fitins_$1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_detect.py 1280 800 256 3 1 > /dev/null 2>&1
