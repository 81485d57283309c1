#!/usr/bin/env python3
"""Copy the rocprofv3 summaries produced by tools/profile_sawyer.sh (gpurun_out/prof_<workload>_*) into profiles/ (tracked):
the kernel-trace stats CSV as is, the SQ counters of the rollout kernel as one JSON (means per launch)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
w = sys.argv[1] if len(sys.argv) > 1 else 'sawyer_peg'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r01'
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(OUT, f'prof_{w}_stats', '*', '*_kernel_stats.csv'))
shutil.copy(stats, os.path.join(PROF, f'{tag}_bench_{w}_kernel_stats.csv'))
kern = [r for r in csv.DictReader(open(stats)) if 'rollout' in r['Name']][0]
res = {'workload': w, 'kernel': kern['Name'], 'launches': int(kern['Calls']), 'mean_ms': float(kern['AverageNs']) / 1e6, 'counters': {}}
# one counter file per pass: the NEWEST in each pass directory (gpurun_out/ keeps the files of earlier rounds next to the new ones)
for f in [newest(os.path.join(d, '*', '*_counter_collection.csv')) for d in sorted(glob.glob(os.path.join(OUT, f'prof_{w}_pmc*'))) if os.path.isdir(d)]:
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(f)):
    if 'rollout' in r['Kernel_Name']:
      agg[r['Counter_Name']].append(float(r['Counter_Value']))
  for c, v in agg.items():
    res['counters'][c] = sum(v) / len(v)
c = res['counters']
if 'SQ_WAVE_CYCLES' in c:
  wc = c['SQ_WAVE_CYCLES']
  res['derived'] = {k: c[n] / wc for k, n in (('issue_any', 'SQ_ACTIVE_INST_ANY'), ('wait_any', 'SQ_WAIT_ANY'), ('wait_inst', 'SQ_WAIT_INST_ANY'),
                                             ('valu', 'SQ_ACTIVE_INST_VALU'), ('lds', 'SQ_ACTIVE_INST_LDS'), ('scalar', 'SQ_ACTIVE_INST_SCA')) if n in c}
  if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
    res['derived']['lds_bank_conflict_share'] = c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1.0)
if 'SQ_THREAD_CYCLES_VALU' in c and 'derived' in res:
  # lanes active per VALU instruction / 64 (SQ_THREAD_CYCLES_VALU counts thread-cycles of VALU work; SQ_INSTS_VALU the instructions)
  res['derived']['lane_occupancy'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_ACTIVE_INST_VALU', 1.0), 1.0))
  res['derived']['lane_occupancy_per_inst'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_INSTS_VALU', 1.0), 1.0))
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
  # HBM bytes per launch as MI355X_MICROARCH.md prescribes for gfx950 (both counters in KiB; FETCH_SIZE reports half the bytes of streaming reads)
  res['hbm_bytes_per_launch'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
  tp = os.path.join(PROF, 'traffic.json')
  tj = json.load(open(tp)) if os.path.exists(tp) else {}
  tj[w] = {"hbm_bytes_per_launch": res["hbm_bytes_per_launch"], "source": f"profiles/{tag}_{w}_rollout_pmc.json", "rocprof_kernel_average_ns": float(kern["AverageNs"]), "issue": res.get("derived"),
           "waves_per_simd": 2 if w == "sawyer_door" else 1}      # the bench batch (8192 envs) runs the door's eight-waves-per-CU build: two waves per SIMD; the peg has one
  json.dump(tj, open(tp, 'w'), indent=1)
json.dump(res, open(os.path.join(PROF, f'{tag}_{w}_rollout_pmc.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
