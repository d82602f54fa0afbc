#!/bin/bash
# kernel timelines of two parts of 8 of the 4K frame, every kernel alone (no side stream): PARTS="0 4" bash tools/tl_two_parts.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export RT_SHADOW_SIDE_SAMPLES=0
for P in ${PARTS:-0 4}; do
  rm -rf $R/gpurun_out/tlq_$P
  timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tlq_$P -o tl -- python3 $R/tools/part_timeline.py $P 8 dragons_4k_ss4_d5 > $R/gpurun_out/tlq_$P.log 2>&1 || { tail -5 $R/gpurun_out/tlq_$P.log; exit 1; }
  python3 $R/tools/frame_timeline.py $(find $R/gpurun_out/tlq_$P -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/tlq_$P.txt
  echo "part $P: $(grep '^ms' $R/gpurun_out/tlq_$P.log | tail -1)"
done
