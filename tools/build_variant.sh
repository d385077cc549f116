#!/bin/bash
# Builds a variant of libvgen_hip.so with extra -D flags for kernels.hip into vgen_amd/libvgen_hip.so.<tag> (A/B runs: tools/ab_fmt.sh)
# usage: bash tools/build_variant.sh tag -DVG_SEQ_WAVES_P2SH=3 ...
set -e
TAG=$1; shift
cd $(dirname $0)/../vgen_amd/csrc
B=../../build/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-unknown-pragmas -I../../include -Wno-pass-failed "$@" -c device/kernels.hip -o /tmp/kernels_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvgen_hip.so.$TAG $B/host/*.o /tmp/kernels_$TAG.o $B/runtime.o $B/cabi.o $B/scanner.o -lpthread -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined
ls -la ../libvgen_hip.so.$TAG
