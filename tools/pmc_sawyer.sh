#!/bin/bash
# Runs on the GPU box (via gpurun): SQ counters of the Sawyer door rollout kernel (separate rocprofv3 passes, kernel-trace only).
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles); ACTIVE_INST_VALU / WAVE_CYCLES = share of wave time issuing VALU.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --workload sawyer_door --steps 2 --warmup 1 --no-cpu"
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf $OUT/pmc_sawyer_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_sawyer_$i -- python3 $ARGS > $OUT/pmc_sawyer_$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT -path '*pmc_sawyer_*' -name '*counter_collection.csv' | head
