#!/bin/bash
# A variant of one kernel unit (ALT_UNIT, default 0: the 8-wave fast-mode task kernel) with extra -D flags, linked with the regular
# objects as kalign_amd/libkalign_amd_<name>.so: A/B runs of compile-time switches (copied over the library on the GPU box).
# usage: [ALT_UNIT=2] tools/build_alt.sh NAME -DKA_W_EARLY=0 ...
set -e
name=$1; shift
cd "$(dirname "$0")/../kalign_amd/csrc"
mkdir -p build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function \
    -DKA_UNIT=${ALT_UNIT:-0} "$@" -c -o build/$name/ka_kernels_u${ALT_UNIT:-0}.o ka_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkalign_amd_$name.so build/$name/ka_kernels_u${ALT_UNIT:-0}.o \
    $(ls build/*.o | grep -v ka_kernels_u${ALT_UNIT:-0}.o)
echo built kalign_amd/libkalign_amd_$name.so
