#!/bin/bash
# timelines: whole 4K frame with level-0 shadow rays on the side stream; part 0/8 with and without the side stream
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; rm -rf $R/gpurun_out/tl_$tag; timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o tl -- python3 $R/tools/part_timeline.py "$@" > $R/gpurun_out/tl_$tag.log 2>&1 || { tail -5 $R/gpurun_out/tl_$tag.log; exit 1; }; python3 $R/tools/frame_timeline.py $(find $R/gpurun_out/tl_$tag -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/tl_$tag.txt; }
export RT_SHADOW_SIDE_SAMPLES=400000000
run whole_side 0 1 dragons_4k_ss4_d5 || exit 1
export RT_SHADOW_SIDE_SAMPLES=0
run part8_noside 0 8 dragons_4k_ss4_d5 || exit 1
run whole_noside 0 1 dragons_4k_ss4_d5 || exit 1
