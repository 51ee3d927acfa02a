#!/bin/bash
# The library with unit 0 (the 8-wave fast-mode task kernel) compiled with -DKA_PROF: per-(level, wave) time stamps of the profiled
# task, cycles inside the paired steps of its strips, its event steps and where they spend their cycles.  The other units are the
# regular objects (build the library first).  On the GPU box: cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so,
# then tools/strip_phases.py.
set -e
cd "$(dirname "$0")/../kalign_amd/csrc"
mkdir -p build/prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function \
    -DKA_UNIT=0 -DKA_PROF -c -o build/prof/ka_kernels_u0.o ka_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkalign_amd_prof.so build/prof/ka_kernels_u0.o \
    $(ls build/*.o | grep -v ka_kernels_u0.o)
echo built kalign_amd/libkalign_amd_prof.so
