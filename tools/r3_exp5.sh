#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 600 python -m pytest tests/test_gpu_dist_c.py -x -q > gpurun_out/r3f/pytest_dist.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r3f/pytest_dist.log
tail -25 gpurun_out/r3f/pytest_dist.log | cut -c1-260
KA_BENCH_FORCE_MULTI=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r3f/force_multi.json 2> gpurun_out/r3f/force_multi.err; echo "force-multi rc $?"
tail -3 gpurun_out/r3f/force_multi.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r3f/force_multi.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step', 'single_gpu_step_ms', 'sharding_overhead_ms')}, d['config']['dist_layer'], d['config']['identical_to_a_single_gpu_run'])
except Exception as e:
    print('no json', e)
PY
timeout 300 python -m pytest tests/test_gpu_dist_emul.py tests/test_gpu_partial.py -x -q 2>&1 | tail -3
