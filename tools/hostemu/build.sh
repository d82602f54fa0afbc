#!/bin/sh
# Host-emulation build of the render core for ASAN/UBSan/gdb (debugging harness; see hip_shim.h).
#   sh tools/hostemu/build.sh            -> build/librtcore_emu.so (sanitized, -O1 -g)
#   LD_PRELOAD=$(gcc -print-file-name=libasan.so) RTCORE_LIB=build/librtcore_emu.so python tools/gpu_parity.py --size=64 csg
set -e
cd "$(dirname "$0")/../.."
mkdir -p build
g++ -std=c++17 -O1 -g -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -DRT_HOST_EMU -I tools/hostemu/fake -include tools/hostemu/hip_shim.h -x c++ raytracer_amd/csrc/rtcore.hip \
    -o build/librtcore_emu.so
echo built build/librtcore_emu.so
