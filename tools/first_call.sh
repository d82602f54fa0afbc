#!/bin/bash
for B in default 67108864 16777216 4194304; do
  if [ $B = default ]; then timeout -k 10 120 python tools/first_call.py 2>&1 | grep call || exit 1
  else RT_STREAM_BATCH_SAMPLES=$B timeout -k 10 120 python tools/first_call.py 2>&1 | grep call || exit 1; fi
done
