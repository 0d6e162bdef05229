#!/bin/bash
# SQ counters of k_tile on the bench-sized threshold+segment workload, two rocprofv3 --pmc passes (run through gpurun);
# tools/tile_pmc_json.py turns the two output directories into profiles/r01_k_tile_pmc.json
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rm -rf $root/gpurun_out/pmc_tile1 $root/gpurun_out/pmc_tile2   # also delete the local copies first: gpurun merges into what is there
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d $root/gpurun_out/pmc_tile1 -- python3 $root/tools/bench_thrseg.py 1280 800 256 synth > /dev/null 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT --output-format csv -d $root/gpurun_out/pmc_tile2 -- python3 $root/tools/bench_thrseg.py 1280 800 256 synth > /dev/null 2>&1 &&
echo pmc done
