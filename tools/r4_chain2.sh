cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 ';KA_CHAIN_TASKS=140;KA_CHAIN_TASKS=125;KA_CHAIN_TASKS=100' 2>&1 | grep -v amdgpu
  VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 ';KA_CHAIN_TASKS=140;KA_CHAIN_TASKS=100;KA_CHAIN_TASKS=64' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 4096 400 0 ';KA_CHAIN_TASKS=200;KA_CHAIN_TASKS=150' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_chain_start_greedy.log 2>&1
cat gpurun_out/r04_chain_start_greedy.log
