#!/bin/bash
# duration of every trace launch of one part of the 4K frame (side stream off), for the libraries in build/ and the part counts given
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export RT_SHADOW_SIDE_SAMPLES=0
for LIB in $R/raytracer_amd/lib/librtcore.so $R/build/librtcore_*.so; do
for NP in ${PARTS:-1 8}; do
  rm -rf $R/gpurun_out/tlp_$NP
  RTCORE_LIB=$LIB timeout -k 10 180 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tlp_$NP -o tl -- python3 $R/tools/part_timeline.py 0 $NP dragons_4k_ss4_d5 > $R/gpurun_out/tlp_$NP.log 2>&1 || { tail -5 $R/gpurun_out/tlp_$NP.log; exit 1; }
  python3 $R/tools/frame_timeline.py $(find $R/gpurun_out/tlp_$NP -name '*kernel_trace.csv' | head -1) > $R/gpurun_out/tlp_$NP.txt
  echo "$(basename $LIB) parts $NP: $(grep -E 'trace_closest|trace_shadow' $R/gpurun_out/tlp_$NP.txt | awk '{printf "%s ", $3}') frame $(grep '^ms' $R/gpurun_out/tlp_$NP.log | tail -1)"
done
done
