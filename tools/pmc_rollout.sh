#!/bin/bash
# SQ counters of the rollout kernel at a given N (GPU box): LDS conflicts, instruction mix, stall buckets.
export TMPDIR=/tmp
N=${1:-16384}
OUT=$PWD/gpurun_out/pmc_$N
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 tools/tune_rollout.py 0 $N > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
  if 'rollout_ws' in r['Kernel_Name']:
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('N=$N', {k: sum(v)/len(v) for k, v in agg.items()})
PY
