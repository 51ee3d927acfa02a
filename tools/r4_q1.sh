#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp kalign_amd/libkalign_amd.so /tmp/reg.so
cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
for per in 1 2 4; do for q in 0 3; do
echo "######## KA_PER=$per KA_Q1=$q"
KA_PER=$per KA_Q1=$q PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids | grep -v "500 x 500" | grep "1000 x\|wave 0\|wave 3" | head -4
done; done
cp /tmp/reg.so kalign_amd/libkalign_amd.so
