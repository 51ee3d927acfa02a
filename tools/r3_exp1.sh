#!/bin/bash
# round 3, GPU call 1: parity suite with the new defaults, then the headline under the new switches
mkdir -p gpurun_out/r3a
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3a/pytest.log
tail -3 gpurun_out/r3a/pytest.log
timeout 500 python tools/variants.py 4096 400 0 'KA_Q1=0,KA_NO_CRIT=1,KA_MAX_CLUSTER=8;KA_Q1=0,KA_NO_CRIT=1;KA_Q1=0;;KA_Q1=2;KA_Q1=3;KA_LEAN4=1;KA_CHAIN_TASKS=150;KA_CHAIN_TASKS=120;KA_CHAIN_TASKS=64;KA_CRIT_TOP=8;KA_CRIT_TOP=2;KA_Q1=0,KA_CHAIN_TASKS=120' > gpurun_out/r3a/variants.log 2>&1
cat gpurun_out/r3a/variants.log
timeout 300 python tools/levels_real.py 0 4096 400 0 x 6,8,10,13,16 > gpurun_out/r3a/levels.log 2>&1
tail -60 gpurun_out/r3a/levels.log
