#!/bin/bash
mkdir -p gpurun_out/r3d
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -x -q > gpurun_out/r3d/pytest_small.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r3d/pytest_small.log
tail -5 gpurun_out/r3d/pytest_small.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -m8 -B2 -A14 "Error\|assert" gpurun_out/r3d/pytest_small.log | cut -c1-250 | head -80; exit 0; fi
timeout 200 python tools/strip_cost.py 2>&1 | tail -8
timeout 400 python tools/variants.py 4096 400 0 ';KA_Q1=1;KA_Q1=2;KA_SUBTREE=0' > gpurun_out/r3d/variants.log 2>&1
cat gpurun_out/r3d/variants.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3d/pytest.log
tail -8 gpurun_out/r3d/pytest.log | cut -c1-300
timeout 300 python tools/levels_real.py 0 4096 400 0 x 8,16 > gpurun_out/r3d/levels.log 2>&1
grep -A14 "^critical\|^task of\|^root task" gpurun_out/r3d/levels.log | cut -c1-200
