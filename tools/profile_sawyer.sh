#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of `bench.py --workload $1` (sawyer_door | sawyer_peg), then SQ
# counter passes of the same command (separate passes, kernel-trace only) -> gpurun_out/prof_$1_*.  tools/summarize_sawyer.py
# copies the summaries into profiles/.
set -u
export TMPDIR=/tmp
W=${1:-sawyer_peg}
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --workload $W --steps 3 --warmup 1 --no-cpu"
rm -rf $OUT/prof_${W}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${W}_stats -- python3 $ARGS > $OUT/prof_${W}_stats.log 2>&1
echo "stats rc=$?"; tail -1 $OUT/prof_${W}_stats.log | cut -c1-160
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do   # the two HBM counters cannot share a pass on gfx950
  i=$((i+1))
  rm -rf $OUT/prof_${W}_pmc$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/prof_${W}_pmc$i -- python3 $ARGS > $OUT/prof_${W}_pmc$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
find $OUT -path "*prof_${W}_*" -name '*.csv' | head -20
