#!/bin/bash
# Round 4, tracked column-record reads (KA_UNTRACKED_READS=0, the default) against the round-3 form (libkalign_amd_untracked.so:
# tools/build_alt.sh untracked -DKA_UNTRACKED_READS=1) and against a build whose helper-wave strips track their reads in the
# steady octets too (libkalign_amd_alltracked.so: -DKA_W_UNTRACKED=0): parity subset on the default build, then the same jobs
# through the three libraries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r04_tracked_reads.log
{
timeout 900 python -m pytest tests/test_gpu_handover.py tests/test_gpu_stress.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
cp kalign_amd/libkalign_amd.so /tmp/reg.so
for lib in reg untracked alltracked; do
  [ $lib != reg ] && { [ -f kalign_amd/libkalign_amd_$lib.so ] || continue; cp kalign_amd/libkalign_amd_$lib.so kalign_amd/libkalign_amd.so; }
  echo "==== library: $lib"
  timeout 400 python tools/variants.py 4096 400 0 'KA_HW=1;KA_HW=0' 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 400 python tools/variants.py 1024 400 0 'KA_HW=1' 2>&1 | tail -1
  timeout 400 python tools/variants.py 1024 2000 1 'KA_HW=1;KA_HW=0' 2>&1 | tail -2
  cp /tmp/reg.so kalign_amd/libkalign_amd.so
done
} > $LOG 2>&1
cat $LOG
