#!/bin/bash
# Build vgen_amd/libvgen_hip.so.<tag> from the in-tree sources with extra compiler flags for device/kernels.hip (A/B of
# compile-time choices; the in-tree library is not touched).  usage: tools/build_flag_variant.sh <tag> -DSOME_SWITCH=1 ...
set -e
TAG=$1; shift
cd "$(dirname "$0")/../vgen_amd/csrc"
make -s device/hash_blocks.inc
D=/tmp/flagvar_$TAG; rm -rf $D; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-pass-failed "$@" -c device/kernels.hip -o $D/kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvgen_hip.so.$TAG ../../build/lib/host/*.o $D/kernels.o ../../build/lib/runtime.o ../../build/lib/cabi.o ../../build/lib/scanner.o -lpthread -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
echo built $TAG
