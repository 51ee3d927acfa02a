#!/bin/bash
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3e/pytest.log
tail -6 gpurun_out/r3e/pytest.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc $?"
tail -3 gpurun_out/r3e/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3e/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'bracket', d.get('value_survey_8d_bracket'), 'roofline', d['roofline']['frac'], 'gaps', d.get('gaps_identical_to_reference'))
print('saturation', [(p['trees_in_flight'], round(p['gcups'], 1)) for p in d['saturation']['points']])
for k in ('c2_1024x400_protein', 'c3_4096x2000_dna', 'c4_single_gpu', 'concurrent_sets', 'refine_all_c2', 'default_mode', 'seqseq_batch'):
    v = d.get(k)
    if isinstance(v, dict): print(k, {x: v[x] for x in v if x in ('value', 'ms_per_step', 'gcups', 'ms_per_round', 'gaps_identical_to_reference', 'device_ms', 'tree_ms', 'gcups_total')})
print('cpu', {x: d['cpu_baseline'][x] for x in ('value', 'cores', 'host_physical_cores', 'gaps_identical_to_reference')})
PY
