#!/bin/bash
# SQ counters of k_tile / k_fmerge on the bench-sized threshold+segment workload, two rocprofv3 --pmc passes (run through gpurun);
# tools/tile_pmc_json.py turns the two output directories into profiles/r02_k_tile_pmc.json.  usage: tools/collect_tile_pmc.sh [kind]
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
kind=${1:-synth}
rm -rf $root/gpurun_out/pmc_tile1 $root/gpurun_out/pmc_tile2   # also delete the local copies first: gpurun merges into what is there
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $root/gpurun_out/pmc_tile1 -- python3 $root/tools/bench_thrseg.py 1280 800 256 $kind > /dev/null 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $root/gpurun_out/pmc_tile2 -- python3 $root/tools/bench_thrseg.py 1280 800 256 $kind > /dev/null 2>&1 &&
echo pmc done
