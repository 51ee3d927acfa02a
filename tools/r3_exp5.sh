#!/bin/bash
mkdir -p gpurun_out/r3f
echo skip
tail -25 gpurun_out/r3f/pytest_dist.log | cut -c1-260
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 KA_BENCH_FORCE_MULTI=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r3f/force_multi.json 2> gpurun_out/r3f/force_multi.err; echo "force-multi rc $?"
tail -3 gpurun_out/r3f/force_multi.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r3f/force_multi.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step', 'single_gpu_step_ms', 'sharding_overhead_ms')}, d['config']['dist_layer'], d['config']['identical_to_a_single_gpu_run'])
except Exception as e:
    print('no json', e)
PY
echo skip2
