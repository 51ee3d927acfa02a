#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 400 python tools/variants.py 4096 400 0 'KA_HW=0;KA_HW=1' 2>&1 | tail -3
timeout 400 python tools/variants.py 1024 400 0 'KA_HW=0;KA_HW=1' 2>&1 | tail -3
for hw in 1 0; do echo "== levels KA_HW=$hw"; KA_HW=$hw timeout 300 python tools/levels_real.py 0 4096 400 2>&1 | grep -A12 "^root task\|critical path (root" | head -40; done
cp kalign_amd/libkalign_amd.so /tmp/reg.so; cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
KA_MAX_CLUSTER=1 PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
cp /tmp/reg.so kalign_amd/libkalign_amd.so
} > gpurun_out/r4_hw2.log 2>&1
tail -120 gpurun_out/r4_hw2.log
