#!/bin/bash
# the kernel-trace pass of tools/profile.sh alone (every kernel alone on the GPU): bash tools/profile_bench_only.sh <tag>
tag=${1:-r03}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export RT_SHADOW_SIDE_SAMPLES=0
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot > $out/bench.log 2>&1 || { echo bench trace failed; tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log > $out/bench_line.json
find $out -name "*kernel_stats.csv" | head -3
