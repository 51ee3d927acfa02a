cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_COPIES=16 VAR_STEPS=3 timeout 600 python tools/variants.py 4096 400 0 'KA_HW=1;KA_HW=0;KA_HW=1,KA_Q1=0;KA_HW=1,KA_NO_HALF=1;KA_HW=1,KA_PER=2;KA_HW=1,KA_MAX_CLUSTER=4' 2>&1 | grep -v amdgpu
  VAR_COPIES=4 VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 'KA_HW=1;KA_HW=0' 2>&1 | grep -v amdgpu
  timeout 300 python tools/refine_time.py 1024 400 2>&1 | grep -v amdgpu | tail -12
  KA_HW=0 timeout 300 python tools/refine_time.py 1024 400 2>&1 | grep -v amdgpu | tail -6
} > gpurun_out/r04_saturation_variants.log 2>&1
cat gpurun_out/r04_saturation_variants.log
