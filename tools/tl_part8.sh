#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/tl8
timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl8 -o tl -- python3 $R/tools/part_timeline.py 0 8 dragons_4k_ss4_d5 > $R/gpurun_out/tl8.log 2>&1 || { tail -5 $R/gpurun_out/tl8.log; exit 1; }
python3 $R/tools/frame_timeline.py $(find $R/gpurun_out/tl8 -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/tl8.txt
grep '^ms' $R/gpurun_out/tl8.log
