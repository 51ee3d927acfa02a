#!/bin/bash
# round 4, first run of the helper-wave strips: A/B against ka_strip on two trees, then the parity tests that exercise them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout 400 python tools/variants.py 1024 400 0 'KA_HW=0;KA_HW=1' 2>&1 | tail -4
timeout 400 python tools/variants.py 4096 400 0 'KA_HW=0;KA_HW=1;KA_HW=1,KA_MAX_CLUSTER=8' 2>&1 | tail -5
timeout 400 python tools/variants.py 1024 2000 1 'KA_HW=0;KA_HW=1' 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_handover.py -x -q 2>&1 | tail -15
} > gpurun_out/r4_hw1.log 2>&1
tail -40 gpurun_out/r4_hw1.log
