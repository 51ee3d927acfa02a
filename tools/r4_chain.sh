#!/bin/bash
# where the chained launch starts (KA_CHAIN_TASKS: at the first level with at most this many tasks) on C3, 1024 x 2000 nt, the headline, C2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 ';KA_CHAIN_TASKS=128;KA_CHAIN_TASKS=100;KA_CHAIN_TASKS=64;KA_CHAIN_TASKS=40;KA_CHAIN_TASKS=24' 2>&1 | grep -v amdgpu
  VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 ';KA_CHAIN_TASKS=128;KA_CHAIN_TASKS=64;KA_CHAIN_TASKS=32;KA_CHAIN_TASKS=16' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 4096 400 0 ';KA_CHAIN_TASKS=128;KA_CHAIN_TASKS=64;KA_CHAIN_TASKS=32' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 1024 400 0 ';KA_CHAIN_TASKS=128;KA_CHAIN_TASKS=64;KA_CHAIN_TASKS=32' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_chain_start.log 2>&1
cat gpurun_out/r04_chain_start.log
