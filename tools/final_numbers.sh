#!/bin/bash
# The round's measured set in one go (run through gpurun): bench line with extras, the two larger BASELINE configs,
# single-frame latency, PCIe-inclusive streaming.  Everything lands under gpurun_out/final/.
out=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $out
python bench.py --steps 10 > $out/bench_default.json 2> $out/bench_default.err && echo "bench done" &&
python tools/bench_latency.py 3 2 > $out/latency_n3_d2.json && python tools/bench_latency.py 3 1 > $out/latency_n3_d1.json &&
python tools/bench_latency.py 1 2 > $out/latency_n1_d2.json && echo "latency done" &&
python tools/bench_stream.py 8 3 1 > $out/stream_n3_d1.json && python tools/bench_stream.py 8 3 2 > $out/stream_n3_d2.json &&
python tools/bench_stream.py 8 1 2 > $out/stream_n1_d2.json && echo "stream done" &&
python bench.py --steps 3 --warmup 1 --width 1920 --height 1080 --batch 512 --tags 30 --unique 64 --no-cpu-baseline --no-extras > $out/bench_c3.json 2> $out/bench_c3.err && echo "c3 done" &&
python bench.py --steps 3 --warmup 1 --width 2448 --height 2048 --batch 256 --tags 20 --unique 32 --no-cpu-baseline --no-extras > $out/bench_c5.json 2> $out/bench_c5.err && echo "c5 done"
