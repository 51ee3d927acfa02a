#!/bin/bash
mkdir -p gpurun_out/r3g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3g/pytest.log
tail -30 gpurun_out/r3g/pytest.log | cut -c1-300
timeout 400 python tools/variants.py 4096 400 0 ';KA_MW=0' > gpurun_out/r3g/variants.log 2>&1
cat gpurun_out/r3g/variants.log
