#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950) -> gpurun_out/prof_*.  Summaries are copied to profiles/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --no-cpu --no-step-api --no-sawyer --no-kitchen --no-minitaur --no-single"   # 20 + 5 launches of 28 episodes each, own actions per episode (the default command = the flags the driver uses)
cd $PWD
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $ARGS > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $ARGS > $OUT/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $ARGS > $OUT/prof_write.log 2>&1
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name '*.csv' | head -30
