cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
echo "== parity sanity"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
echo "== saturation 16 trees"; VAR_COPIES=16 VAR_STEPS=3 timeout 900 python tools/variants.py 4096 400 0 ';KA_QW=2;KA_QW=1;KA_LW=2;KA_LW=1;KA_QW=2,KA_LW=2;KA_QW=1,KA_LW=1;KA_QW=1,KA_LW=2' 2>&1 | grep -v amdgpu
echo "== headline single tree"; VAR_STEPS=8 timeout 600 python tools/variants.py 4096 400 0 ';KA_QW=2;KA_QW=1;KA_LW=2;KA_LW=1;KA_QW=1,KA_LW=1' 2>&1 | grep -v amdgpu
echo "== pairs 4096x400 K=5"; VAR_PAIRS=5 timeout 600 python tools/variants.py 4096 400 0 ';KA_PW=2;KA_PW=1' 2>&1 | grep -v amdgpu
echo "== default-mode tree 4096x400"; VAR_ANCHORS=5 VAR_STEPS=4 timeout 600 python tools/variants.py 4096 400 0 ';KA_QW=2;KA_QW=1;KA_LW=2;KA_LW=1' 2>&1 | grep -v amdgpu
echo "== dropin multi + adopt"; timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -k "several_ranks or inline_refinement" 2>&1 | tail -5
} > gpurun_out/r05_exp1.log 2>&1
cat gpurun_out/r05_exp1.log
