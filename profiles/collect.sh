#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun):  bash profiles/collect.sh <tag> [workloads...]
# For every workload (headline = bench.py's default 4096 x 400; c2 = 1024 x 400; c3 = 4096 DNA x 2000): one
# `rocprofv3 --kernel-trace --stats` run of `bench.py --steps 10 --warmup 3 --no-legs --no-cpu`, then SEPARATE `--pmc`
# passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass on gfx950) and, for the headline, the SQ issue counters.
# Counters are never combined with trace domains.  profiles/summarize.py folds the outputs into profiles/<tag>_*.
set -u
TAG=${1:-r02}
shift
WLS=${@:-headline c2 c3}
ROOT=$(pwd)
export TMPDIR=/tmp
for WL in $WLS; do
  if [ $WL = refine ]; then
    # refinement (ka_tree_refine, modes 4 / 1 / 2 / 3 after a first pass) on 1024 x 400: kernel trace only
    OUT=$ROOT/gpurun_out/prof_${TAG}_refine
    mkdir -p $OUT
    cd /tmp
    timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/tools/refine_time.py 1024 400 > $OUT/kt.log 2>&1
    cd $ROOT
    cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
    tail -12 $OUT/kt.log
    continue
  fi
  case $WL in
    headline) ARGS="";;
    c2) ARGS="--nseq 1024 --len 400";;
    c3) ARGS="--nseq 4096 --len 2000 --dna";;
  esac
  OUT=$ROOT/gpurun_out/prof_${TAG}_$WL
  mkdir -p $OUT
  python bench.py --steps 10 --warmup 3 --no-legs --no-cpu $ARGS > $OUT/bench.json 2> $OUT/bench.err
  cd /tmp
  timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-legs --no-cpu $ARGS > $OUT/kt.log 2>&1
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --no-legs --no-cpu $ARGS > $OUT/pmc_fetch.log 2>&1
  timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py --steps 3 --warmup 1 --no-legs --no-cpu $ARGS > $OUT/pmc_write.log 2>&1
  if [ $WL = headline ]; then
    timeout -s KILL 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py --steps 3 --warmup 1 --no-legs --no-cpu $ARGS > $OUT/pmc_sq.log 2>&1
  fi
  cd $ROOT
  python profiles/summarize.py $TAG $WL $OUT > $OUT/summary.log 2>&1
  cat $OUT/summary.log
done
