#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/profile_kitchen.sh / profile_minitaur.sh into profiles/ (kernel stats CSV; SQ counters of the fused rollout
kernel as JSON) and record the issue shares in profiles/traffic.json.   usage: summarize_kitchen.py [tag] [kitchen|minitaur]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
w = sys.argv[2] if len(sys.argv) > 2 else 'kitchen'
KERNEL = {'kitchen': 'kitchen_rollout_kernel', 'minitaur': 'minitaur_'}[w]      # (minitaur: minitaur_kernel<..> or minitaur_duo_kernel, whichever the launch took)      # (kitchen: the fused rollout; its step_api leg launches physics_kernel<23, 32>)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(OUT, f'prof_{w}_stats', '*', '*_kernel_stats.csv'))
shutil.copy(stats, os.path.join(PROF, f'{tag}_bench_{w}_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
kerns = [r for r in rows if KERNEL in r['Name']]
kern = max(kerns, key=lambda r: float(r['TotalDurationNs']) if 'TotalDurationNs' in r else float(r['AverageNs']))     # (minitaur: the rollout instantiation, not the reset one)
res = {'workload': w, 'kernel': kern['Name'], 'launches': int(kern['Calls']), 'mean_ms': float(kern['AverageNs']) / 1e6,
       'share_of_gpu_time': float(kern['Percentage']), 'counters': {}}
for d in sorted(glob.glob(os.path.join(OUT, f'prof_{w}_pmc*'))):
  if not os.path.isdir(d):
    continue
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(newest(os.path.join(d, '*', '*_counter_collection.csv')))):
    if r['Kernel_Name'] == kern['Name'] or (w == 'kitchen' and KERNEL in r['Kernel_Name']):
      agg[r['Counter_Name']].append(float(r['Counter_Value']))
  for c, v in agg.items():
    res['counters'][c] = sum(v) / len(v)
c = res['counters']
if 'SQ_WAVE_CYCLES' in c:
  wc = c['SQ_WAVE_CYCLES']
  res['derived'] = {k: c[n] / wc for k, n in (('issue_any', 'SQ_ACTIVE_INST_ANY'), ('wait_any', 'SQ_WAIT_ANY'), ('wait_inst', 'SQ_WAIT_INST_ANY'),
                                             ('valu', 'SQ_ACTIVE_INST_VALU'), ('lds', 'SQ_ACTIVE_INST_LDS'), ('scalar', 'SQ_ACTIVE_INST_SCA')) if n in c}
  if 'SQ_THREAD_CYCLES_VALU' in c:
    # lanes active per VALU instruction / 64 (SQ_THREAD_CYCLES_VALU counts thread-cycles of VALU work)
    res['derived']['lane_occupancy'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_ACTIVE_INST_VALU', 1.0), 1.0))
    res['derived']['lane_occupancy_per_inst'] = c['SQ_THREAD_CYCLES_VALU'] / (64.0 * max(c.get('SQ_INSTS_VALU', 1.0), 1.0))
json.dump(res, open(os.path.join(PROF, f'{tag}_{w}_rollout_pmc.json'), 'w'), indent=1)
tp = os.path.join(PROF, 'traffic.json')
tj = json.load(open(tp)) if os.path.exists(tp) else {}
if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:      # HBM bytes per launch as MI355X_MICROARCH.md prescribes for gfx950 (both counters in KiB; FETCH_SIZE reports half the bytes of streaming reads)
  res['hbm_bytes_per_launch'] = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024
  json.dump(res, open(os.path.join(PROF, f'{tag}_{w}_rollout_pmc.json'), 'w'), indent=1)
tj[w] = {'hbm_bytes_per_launch': res.get('hbm_bytes_per_launch'), 'source': f'profiles/{tag}_{w}_rollout_pmc.json', 'rocprof_kernel_average_ns': float(kern['AverageNs']), 'issue': res.get('derived'), 'waves_per_simd': 2 if 'duo' in kern['Name'] else 1}
json.dump(tj, open(tp, 'w'), indent=1)
print(json.dumps(res, indent=1))
