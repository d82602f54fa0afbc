#!/bin/bash
# bench lines of the production library and of every build/librtcore_*.so
AB_WORKLOADS=${AB_WORKLOADS:-dragons_4k_ss4_d5} timeout -k 10 900 python tools/ab_variants.py raytracer_amd/lib/librtcore.so build/librtcore_*.so
