#!/bin/bash
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c/pytest.log
tail -15 gpurun_out/r3c/pytest.log
timeout 400 python tools/variants.py 4096 400 0 'KA_SUBTREE=0;;KA_SUBTREE=0,KA_LEAN4=0' > gpurun_out/r3c/variants.log 2>&1
cat gpurun_out/r3c/variants.log
timeout 300 python tools/levels_real.py 0 4096 400 0 x 8,16 > gpurun_out/r3c/levels.log 2>&1
tail -75 gpurun_out/r3c/levels.log | cut -c1-180
