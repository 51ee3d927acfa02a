#!/bin/bash
# A/B of library builds on the GPU box: the regular library, kalign_amd/libkalign_amd_alt.so (tools/build_alt.sh) and the KA_PROF
# build (tools/build_prof.sh) through the same jobs.  usage: tools/r4_ab.sh [log name]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/${1:-r4_ab}.log
cp kalign_amd/libkalign_amd.so /tmp/reg.so
{
for lib in reg alt; do
  [ $lib = alt ] && { [ -f kalign_amd/libkalign_amd_alt.so ] || continue; cp kalign_amd/libkalign_amd_alt.so kalign_amd/libkalign_amd.so; }
  echo "==== library: $lib"
  timeout 400 python tools/variants.py 4096 400 0 "${AB_VARIANTS:-KA_HW=1;KA_HW=0}" 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 400 python tools/variants.py 1024 400 0 'KA_HW=1' 2>&1 | tail -1
  timeout 400 python tools/variants.py 1024 2000 1 'KA_HW=1' 2>&1 | tail -1
  cp /tmp/reg.so kalign_amd/libkalign_amd.so
done
echo "== levels KA_HW=1"; timeout 300 python tools/levels_real.py 0 4096 400 2>&1 | grep -A22 "^root task\|critical path (root" | head -60
cp kalign_amd/libkalign_amd_prof.so kalign_amd/libkalign_amd.so
PHASES_HW=1 timeout 300 python tools/strip_phases.py 2>&1 | grep -v amdgpu.ids
cp /tmp/reg.so kalign_amd/libkalign_amd.so
} > $LOG 2>&1
cat $LOG
