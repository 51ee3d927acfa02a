#!/bin/bash
# workgroups a join has no room for wait in a pool for a later join (KA_POOL=1, default) against leaving (0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
  VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 'KA_POOL=1;KA_POOL=0;KA_POOL=1,KA_CRIT_GREEDY=0' 2>&1 | grep -v amdgpu
  VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 'KA_POOL=1;KA_POOL=0' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 4096 400 0 'KA_POOL=1;KA_POOL=0' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 1024 400 0 'KA_POOL=1;KA_POOL=0' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 2048 1000 0 'KA_POOL=1;KA_POOL=0' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_pool.log 2>&1
cat gpurun_out/r04_pool.log
