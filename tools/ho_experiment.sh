#!/bin/bash
# LDS hand-over (KA_HO) experiment: A/B timings on the headline and a DNA job, parity tests.  Run on the GPU box from the repo root.
mkdir -p gpurun_out
export VAR_STEPS=8
timeout 400 python tools/variants.py 4096 400 0 ';KA_HO=1;KA_HO=2;KA_HO=1,KA_Q1=1;KA_HO=2,KA_Q1=1;KA_HO=2,KA_Q1=2;KA_HO=2,KA_Q1=3;KA_Q1=1;' > gpurun_out/ho_headline.log 2>&1
echo "headline rc=$?"; cat gpurun_out/ho_headline.log
timeout 400 python tools/variants.py 1024 2000 1 ';KA_HO=1;KA_HO=2;KA_HO=2,KA_Q1=1' > gpurun_out/ho_dna.log 2>&1
echo "dna rc=$?"; cat gpurun_out/ho_dna.log
timeout 600 python -m pytest tests/test_gpu_handover.py -x -q > gpurun_out/ho_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/ho_tests.log
KA_STRESS_REPS=10 timeout 600 python -m pytest tests/test_gpu_stress.py -x -q > gpurun_out/ho_stress.log 2>&1
echo "stress rc=$?"; tail -15 gpurun_out/ho_stress.log
