#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun):  bash profiles/collect.sh <tag>
# Produces under gpurun_out/prof_<tag>/ : bench JSON, rocprofv3 kernel-trace stats, two PMC passes
# (FETCH_SIZE and WRITE_SIZE separately -- they do not fit one pass on gfx950), then
# profiles/summarize.py folds them into profiles/<tag>_*.  Counters are never combined with trace domains.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-pairs > $OUT/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pairs > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pairs > $OUT/pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pairs > $OUT/pmc_sq.log 2>&1
cd $ROOT
python profiles/summarize.py $TAG $OUT > $OUT/summary.log 2>&1
cat $OUT/summary.log
