#!/bin/bash
# spare chain workgroups by the simulated schedule (KA_CRIT_GREEDY=1, default) against the ranking alone (0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ python tools/plan_dump.py 2>&1 | grep -v amdgpu | grep -A30 "greedy pass" | head -24
  VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 'KA_CRIT_GREEDY=1;KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
  VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 'KA_CRIT_GREEDY=1;KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 4096 400 0 'KA_CRIT_GREEDY=1;KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 1024 400 0 'KA_CRIT_GREEDY=1;KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 2048 1000 0 'KA_CRIT_GREEDY=1;KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_crit_greedy.log 2>&1
cat gpurun_out/r04_crit_greedy.log
