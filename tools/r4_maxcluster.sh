cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_STEPS=3 timeout 600 python tools/variants.py 4096 2000 1 ';KA_MAX_CLUSTER=24;KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
  VAR_STEPS=4 timeout 600 python tools/variants.py 1024 2000 1 ';KA_MAX_CLUSTER=24;KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
  timeout 600 python tools/variants.py 4096 400 0 ';KA_MAX_CLUSTER=24;KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 2048 1000 0 ';KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
} > gpurun_out/r04_max_cluster.log 2>&1
cat gpurun_out/r04_max_cluster.log
