cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 'KA_HW=1;KA_Q1=4;KA_Q1=4,KA_MAX_CLUSTER=8;KA_Q1=1' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 'KA_HW=1;KA_Q1=4' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_q1_dna.log 2>&1
cat gpurun_out/r04_q1_dna.log
