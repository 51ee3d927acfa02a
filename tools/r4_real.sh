#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cp kalign_amd/libkalign_amd.so /tmp/reg.so
cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
{
PHASES_REAL=1 PHASES_LEVELS=3 PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
KA_PER=4 PHASES_REAL=1 PHASES_LEVELS=1 PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
KA_PER=2 PHASES_REAL=1 PHASES_LEVELS=1 PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/${1:-r4_real}.log 2>&1
cp /tmp/reg.so kalign_amd/libkalign_amd.so
cat gpurun_out/${1:-r4_real}.log
