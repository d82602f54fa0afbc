#!/bin/bash
# times tools/small_frames.py with every build/librtcore_*.so (and the production library first)
echo "variant production"
timeout -k 10 200 python tools/small_frames.py > gpurun_out/var_production.log 2>&1 || { echo failed; tail -5 gpurun_out/var_production.log; exit 1; }
grep part gpurun_out/var_production.log
for F in build/librtcore_*.so; do
  V=$(basename $F .so); V=${V#librtcore_}
  echo "variant $V"
  RTCORE_LIB=$PWD/$F timeout -k 10 200 python tools/small_frames.py > gpurun_out/var_$V.log 2>&1 || { echo "failed at $V"; tail -5 gpurun_out/var_$V.log; exit 1; }
  grep part gpurun_out/var_$V.log
done
