#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of `bench.py --workload kitchen`, then SQ counter passes of the same command
# -> gpurun_out/prof_kitchen_*.  tools/summarize_kitchen.py copies the summaries into profiles/.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --workload kitchen --steps 1 --warmup 1 --no-cpu --no-step-api"
rm -rf $OUT/prof_kitchen_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kitchen_stats -- python3 $ARGS > $OUT/prof_kitchen_stats.log 2>&1
echo "stats rc=$?"; tail -1 $OUT/prof_kitchen_stats.log | cut -c1-160
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rm -rf $OUT/prof_kitchen_pmc$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_kitchen_pmc$i -- python3 $ARGS > $OUT/prof_kitchen_pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
