cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
cp kalign_amd/libkalign_amd.so /tmp/reg.so
{
for lib in reg rot rurot; do
  [ $lib != reg ] && cp kalign_amd/libkalign_amd_$lib.so kalign_amd/libkalign_amd.so
  echo "==== library: $lib"
  VAR_COPIES=16 VAR_STEPS=3 timeout 900 python tools/variants.py 4096 400 0 ';KA_QW=2' 2>&1 | grep -v amdgpu
  cp /tmp/reg.so kalign_amd/libkalign_amd.so
done
echo "== dropin: several ranks, adoption, ensemble members side by side"; timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -k "several_ranks or inline_refinement" 2>&1 | tail -15
} > gpurun_out/r05_exp5.log 2>&1
cat gpurun_out/r05_exp5.log
