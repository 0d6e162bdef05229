#!/bin/bash
# Round 4's evidence set in one gpurun call: the bench line, per-kernel durations of the bench batch and of BASELINE configs 3 and 5, the two
# PMC passes behind roofline.traffic, k_tile's counters, the roofline stage alone on dense and on the bench's low-noise frames, k_tile by phase.
# usage (through gpurun): tools/r4_final.sh <tag>     (outputs under gpurun_out/<tag>*)
tag=${1:-r04f}
root=$GRAFT_REPO_ROOT
cd $root
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo "bench done" &&
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1 && echo "profiles done" &&
bash tools/collect_tile_pmc.sh synth > gpurun_out/${tag}_tilepmc.log 2>&1 && echo "tile pmc done" &&
bash tools/prof_thrseg_kernels.sh ${tag}_dense synth > gpurun_out/${tag}_thrseg_dense.txt 2>&1 &&
bash tools/prof_thrseg_kernels.sh ${tag}_bench bench > gpurun_out/${tag}_thrseg_bench.txt 2>&1 &&
bash tools/prof_thrseg_kernels.sh ${tag}_quiet bench_quiet > gpurun_out/${tag}_thrseg_quiet.txt 2>&1 && echo "thrseg done" &&
bash tools/tile_valu_by_phase.sh synth > gpurun_out/${tag}_valu_by_phase.txt 2>&1 && echo "phases done"
