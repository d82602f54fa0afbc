#!/bin/bash
# bench lines of every build/librtcore_*.so, then the statistics of the production library
AB_WORKLOADS=${AB_WORKLOADS:-dragons_4k_ss4_d5} timeout -k 10 900 python tools/ab_variants.py build/librtcore_*.so || exit 1
timeout -k 10 200 python tools/stats_stream.py dragons_4k_ss4_d5 2>&1 | grep -E "^  (shadow|closest)"
