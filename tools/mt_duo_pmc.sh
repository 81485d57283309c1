#!/bin/bash
# SQ counters of the minitaur rollout kernel, one-wave form (0) and two-waves-per-SIMD form (1): is the CU's LDS pipe what the second wave per SIMD runs into?
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/mt_duo_pmc
rm -rf $OUT; mkdir -p $OUT
for mode in 0 1; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/m${mode}_p$i -- python3 tools/mt_duo_run.py $mode 4096 60 > $OUT/m${mode}_p$i.log 2>&1
    echo "mode $mode pass $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), 'gpurun_out', 'mt_duo_pmc')
for mode in (0, 1):
  agg = collections.defaultdict(list)
  for f in glob.glob(os.path.join(out, f'm{mode}_p*', '*', '*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
      if 'minitaur' in r['Kernel_Name'] and ('duo' in r['Kernel_Name'] or 'ILb0' in r['Kernel_Name'] or '<false' in r['Kernel_Name']):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
  c = {k: sum(v) / len(v) for k, v in agg.items()}
  print('mode', mode, {k: f'{v:.4g}' for k, v in sorted(c.items())})
  if 'SQ_WAVE_CYCLES' in c:
    wc = c['SQ_WAVE_CYCLES']
    print('   per wave cycle:', {k: round(c[k] / wc, 4) for k in c if k.startswith('SQ_ACTIVE') or k.startswith('SQ_WAIT') or k.startswith('SQ_LDS')})
    if 'SQ_BUSY_CYCLES' in c: print('   wave cycles / busy cycles', wc / c['SQ_BUSY_CYCLES'])
PY
