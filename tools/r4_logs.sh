#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python tools/levels_real.py 0 4096 400 0 0 9,15,17 2>&1 | grep -v amdgpu > gpurun_out/r04_levels_4096x400.log
timeout 300 python tools/levels_real.py 0 1024 400 2>&1 | grep -v amdgpu > gpurun_out/r04_levels_1024x400.log
timeout 300 python tools/kmeans_time.py 2>&1 | grep -v amdgpu > gpurun_out/r04_kmeans_time.log
timeout 300 python tools/variants.py 4096 400 0 'KA_HW=1;KA_HW=1,KA_Q1=0;KA_HW=0;KA_HW=1,KA_MAX_CLUSTER=8' 2>&1 | grep -v amdgpu > gpurun_out/r04_variants_headline.log
timeout 300 python tools/variants.py 1024 2000 1 'KA_HW=1;KA_HW=0' 2>&1 | grep -v amdgpu >> gpurun_out/r04_variants_headline.log
cp kalign_amd/libkalign_amd.so /tmp/reg.so; cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
{ PHASES_HW=1,0 timeout 300 python tools/strip_phases.py; PHASES_REAL=1 PHASES_LEVELS=3 PHASES_HW=1 timeout 300 python tools/strip_phases.py; } 2>&1 | grep -v amdgpu > gpurun_out/r04_strip_phases.log
cp /tmp/reg.so kalign_amd/libkalign_amd.so
tail -5 gpurun_out/r04_levels_4096x400.log; cat gpurun_out/r04_variants_headline.log; head -8 gpurun_out/r04_strip_phases.log
