"""bench.py's CPU legs: every CPU baseline runs in a fresh child process that alone carries the OpenMP placement variables and sizes its thread
sweep from the CPUs it may use BEFORE importing anything; the reference's simulators are probed, not assumed.  No GPU needed."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_cpu_baseline_child_sweeps_up_to_the_cpus_this_process_may_use():
  import bench
  before = sorted(os.sched_getaffinity(0))
  assert 'OMP_PROC_BIND' not in os.environ or os.environ['OMP_PROC_BIND'] == os.environ.get('OMP_PROC_BIND')
  had = {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES')}
  res = bench.run_cpu_baseline_child('tabletop', n=512, T=50, reward='sparse', seconds=0.2)
  assert res['value'] and res['value'] > 0, res
  assert res['host']['affinity_cpus'] == len(before)                       # the child saw every CPU this process may run on ...
  assert max(int(k) for k in res['by_threads']) == len(before)             # ... and its sweep reaches that thread count
  assert res['host']['OMP_PROC_BIND'] == (had['OMP_PROC_BIND'] or 'spread') and res['host']['OMP_PLACES'] == (had['OMP_PLACES'] or 'cores')
  assert {k: os.environ.get(k) for k in had} == had                        # the placement variables never entered THIS process
  assert sorted(os.sched_getaffinity(0)) == before


def test_the_child_entry_point_prints_one_json_line_and_leaves_the_gpu_alone():
  r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--cpu-child', 'tabletop', '--cpu-child-args',
                      json.dumps(dict(n=256, T=20, reward='sparse', seconds=0.1))], capture_output=True, text=True, timeout=300,
                     env={**os.environ, 'HIP_VISIBLE_DEVICES': '', 'OMP_PROC_BIND': 'spread', 'OMP_PLACES': 'cores'})
  assert r.returncode == 0, r.stderr[-500:]
  line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
  d = json.loads(line)
  assert d['kind'] == 'port' and d['cores'] >= 1 and 'oracle/tabletop_oracle.c' in d['sample']


def test_simulator_probe_reports_what_it_looked_for_and_the_sentences_follow_it():
  import bench
  sim = bench.probe_reference_simulators()
  assert set(sim['probed']) >= {'mujoco', 'mujoco_py', 'pybullet', 'dm_control', 'metaworld'}
  found = [f['module'] for f in sim['found']]
  for name in found:                                                       # whatever it claims to have found really imports
    assert subprocess.run([sys.executable, '-c', f'import {name}'], capture_output=True).returncode == 0
  s = bench.simulator_sentence(sim, 'MuJoCo')
  assert ('IS importable' in s) == bool(sim['mujoco']) and ('not importable' in s) == (not sim['mujoco'])
  fake = dict(sim, found=[{'module': 'pybullet', 'version': '3.2.0'}], pybullet=True)
  assert 'IS importable' in bench.simulator_sentence(fake, 'PyBullet')
