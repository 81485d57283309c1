#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/profile_kitchen.sh into profiles/ (kernel stats CSV; SQ counters of the fused nv = 23 rollout kernel as JSON)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(OUT, 'prof_kitchen_stats', '*', '*_kernel_stats.csv'))
shutil.copy(stats, os.path.join(PROF, f'{tag}_bench_kitchen_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
kern = [r for r in rows if 'kitchen_rollout_kernel' in r['Name']][0]     # the fused rollout (bench.py --workload kitchen); its step_api leg launches physics_kernel<23, 32>
res = {'workload': 'kitchen', 'kernel': kern['Name'], 'launches': int(kern['Calls']), 'mean_ms': float(kern['AverageNs']) / 1e6,
       'share_of_gpu_time': float(kern['Percentage']), 'counters': {}}
for d in sorted(glob.glob(os.path.join(OUT, 'prof_kitchen_pmc*'))):
  if not os.path.isdir(d):
    continue
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(newest(os.path.join(d, '*', '*_counter_collection.csv')))):
    if 'kitchen_rollout_kernel' in r['Kernel_Name']:
      agg[r['Counter_Name']].append(float(r['Counter_Value']))
  for c, v in agg.items():
    res['counters'][c] = sum(v) / len(v)
c = res['counters']
if 'SQ_WAVE_CYCLES' in c:
  wc = c['SQ_WAVE_CYCLES']
  res['derived'] = {k: c[n] / wc for k, n in (('issue_any', 'SQ_ACTIVE_INST_ANY'), ('wait_any', 'SQ_WAIT_ANY'), ('wait_inst', 'SQ_WAIT_INST_ANY'),
                                             ('valu', 'SQ_ACTIVE_INST_VALU'), ('lds', 'SQ_ACTIVE_INST_LDS'), ('scalar', 'SQ_ACTIVE_INST_SCA')) if n in c}
if 'SQ_THREAD_CYCLES_VALU' in c and 'derived' in res:
  # lanes active per VALU instruction / 64 (SQ_THREAD_CYCLES_VALU counts thread-cycles of VALU work; SQ_INSTS_VALU the instructions)
  res['derived']['lane_occupancy'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_ACTIVE_INST_VALU', 1.0), 1.0))
  res['derived']['lane_occupancy_per_inst'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_INSTS_VALU', 1.0), 1.0))
json.dump(res, open(os.path.join(PROF, f'{tag}_kitchen_rollout_pmc.json'), 'w'), indent=1)
tp = os.path.join(PROF, 'traffic.json')
tj = json.load(open(tp)) if os.path.exists(tp) else {}
tj['kitchen'] = {'source': f'profiles/{tag}_kitchen_rollout_pmc.json', 'rocprof_kernel_average_ns': float(kern['AverageNs']), 'issue': res.get('derived'), 'waves_per_simd': 1}
json.dump(tj, open(tp, 'w'), indent=1)
print(json.dumps(res, indent=1))
