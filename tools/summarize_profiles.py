#!/usr/bin/env python3
"""Copy the rocprofv3 summaries produced by tools/profile_bench.sh (gpurun_out/prof_*) into profiles/ (tracked) and
derive the per-launch HBM traffic of the rollout kernel from the PMC passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming
read -> doubled; WRITE_SIZE is exact for streaming stores.  Both counters are in KiB.  Calibration on this access
pattern: the action tensor read by one launch is exactly n*T*12 B, see "calibration" in the JSON."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
PROF = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
n, T, E = 4096, 200, 28         # bench.py defaults: 28 evaluation episodes per launch (four groups of seven in flight)
SUF = f'_E{E}_own_actions' if E > 1 else ''
os.makedirs(PROF, exist_ok=True)

newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
stats = newest(os.path.join(OUT, 'prof_stats', '*', '*_kernel_stats.csv'))
shutil.copy(stats, os.path.join(PROF, f'{tag}_bench_n{n}_T{T}{SUF}_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))
kern = [r for r in rows if 'rollout' in r['Name']][0]

pmc = {}
for name in ('fetch', 'write'):
  f = newest(os.path.join(OUT, f'prof_{name}', '*', '*_counter_collection.csv'))
  agg = collections.defaultdict(list)
  for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
  for (k, c), v in agg.items():
    if 'rollout' in k or 'reset_kernel' in k or 'step_kernel' in k:
      pmc.setdefault(k, {})[c] = {'launches': len(v), 'mean_KiB': sum(v) / len(v), 'min_KiB': min(v), 'max_KiB': max(v)}
rk = [k for k in pmc if 'rollout' in k][0]
fetch, write = pmc[rk]['FETCH_SIZE']['mean_KiB'], pmc[rk]['WRITE_SIZE']['mean_KiB']
hbm = (2 * fetch + write) * 1024
algo = n * (E * T * 66 + 2 * (32 + 1 + 4) + 4)
summary = {
    'command': 'python3 bench.py --no-cpu --no-step-api --no-sawyer --no-kitchen --no-single  [defaults: --steps 20 --warmup 5 = 25 launches of 28 evaluation episodes each, every episode with its own actions: the launch shape does not depend on the flags]  (under rocprofv3, see tools/profile_bench.sh)',
    'kernel': rk,
    'kernel_trace_stats': {'calls': int(kern['Calls']), 'average_ns': float(kern['AverageNs']), 'min_ns': float(kern['MinNs']),
                           'max_ns': float(kern['MaxNs']), 'stddev_ns': float(kern['StdDev'])},
    'pmc': pmc,
    'corrections': 'HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE reads 1/2 for streaming loads)',
    'calibration': {'action_bytes_read_per_launch': n * T * 12 * E, 'FETCH_SIZE_x2_bytes': 2 * fetch * 1024,
                    'note': 'every episode of the launch reads its own actions (act_episode_stride = T*n*3); the remaining read bytes are the fp64 state, goal rows and flags (<= 0.4 MB)'},
    'hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': algo, 'traffic_over_algorithmic': hbm / algo,
}
json.dump(summary, open(os.path.join(PROF, f'{tag}_bench_n{n}_T{T}{SUF}_pmc.json'), 'w'), indent=1)
traffic_path = os.path.join(PROF, 'traffic.json')
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
traffic[f'eval_n{n}_T{T}{SUF}' if E > 1 else f'rollout_n{n}_T{T}'] = {'hbm_bytes_per_launch': hbm, 'source': f'profiles/{tag}_bench_n{n}_T{T}{SUF}_pmc.json',
                                 'rocprof_kernel_average_ns': float(kern['AverageNs'])}
json.dump(traffic, open(traffic_path, 'w'), indent=1)
print(json.dumps({k: summary[k] for k in ('kernel_trace_stats', 'hbm_bytes_per_launch', 'algorithmic_bytes_per_launch', 'traffic_over_algorithmic')}, indent=1))
