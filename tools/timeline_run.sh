#!/bin/bash
# kernel-trace timeline of the last frame of a workload: tools/timeline_run.sh <workload> [part n_parts]
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=$1; PART=${2:-0}; NP=${3:-1}
OUT=$R/gpurun_out/tl_$WL
rm -rf $OUT
timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python3 $R/tools/part_timeline.py $PART $NP $WL > $OUT.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 $R/tools/frame_timeline.py $F > $R/gpurun_out/tl_$WL.txt
grep '^ms' $OUT.log | tail -2
