#!/bin/bash
# SIMD utilisation per kind of traversal step (dev builds with RT_COUNT_PHASE = 1, 2, 3 in build/)
timeout -k 10 200 python tools/stats_stream.py dragons_4k_ss4_d5 2>&1 | grep -E "shadow|closest" | grep -v "^dragons"
for V in phase1 phase2 phase3; do
  echo "== $V"
  RTCORE_LIB=$PWD/build/librtcore_$V.so timeout -k 10 200 python tools/stats_stream.py dragons_4k_ss4_d5 2>&1 | grep -E "shadow|closest" | grep -v "^dragons" || exit 1
done
