#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp kalign_amd/libkalign_amd.so /tmp/reg.so
cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
KA_Q1=1 PHASES_REAL=1 PHASES_LEVELS=4 PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
cp /tmp/reg.so kalign_amd/libkalign_amd.so
