cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{ VAR_STEPS=3 timeout 900 python tools/variants.py 16384 500 0 ';KA_MAX_CLUSTER=32;KA_MAX_CLUSTER=24' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 8192 300 0 ';KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
  VAR_STEPS=3 timeout 600 python tools/variants.py 512 3000 1 ';KA_MAX_CLUSTER=32' 2>&1 | grep -v amdgpu
} >> gpurun_out/r04_max_cluster.log 2>&1
tail -8 gpurun_out/r04_max_cluster.log
