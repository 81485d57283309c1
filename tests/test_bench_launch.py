"""`python bench.py --gpus 2` started the way the driver starts it -- no launcher around it -- must launch itself: main() starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py <same arguments>` as a child process, relays rank 0's ONE JSON line and exits with the
child's code.  No GPU here: the ranks run over gloo on host tensors with the env of tests/bench_stub_env.py (the library's argument validator + synthetic outputs),
every side leg switched off; everything else -- argument hand-over, rendezvous on 127.0.0.1, barrier / timed loop / the job's one all-gather / MAX-reduce, the line
-- is bench.py's own code."""
import json
import os
import subprocess
import sys

from conftest import REPO

ARGS = ['--steps', '3', '--warmup', '1', '--envs', '48', '--horizon', '7', '--episodes-per-launch', '3', '--action-sets', '2', '--settle-launches', '0',
        '--roofline-windows', '3', '--no-cpu', '--no-step-api', '--no-single', '--no-sawyer', '--no-kitchen', '--no-minitaur',
        '--test-env-factory', 'tests.bench_stub_env:make', '--test-backend', 'gloo']


def _run(gpus, extra_env=None):
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  env.update(extra_env or {})
  return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(gpus), *ARGS], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)


LINE_KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline',
             'cpu_baseline', 'full'}
ROOFLINE_KEYS = {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'frac_min', 'frac_max', 'kernel_ms_median', 'strict_frac'}


def _the_line(r):
  """the LAST stdout line is the one JSON line, compact (the driver could not parse the 20 kB line of round 5)"""
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout
  assert len(lines[0]) < 4096
  d = json.loads(lines[0])
  assert set(d) == LINE_KEYS and set(d['roofline']) >= ROOFLINE_KEYS, (sorted(d), sorted(d['roofline']))
  full = [ln for ln in r.stderr.splitlines() if ln.startswith('bench_full: ')]
  assert len(full) == 1
  return d, json.loads(full[0][len('bench_full: '):])


def test_gpus_2_launches_itself_and_prints_one_line():
  r = _run(2)
  d, full = _the_line(r)
  assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
  assert d['config']['global_envs'] == 96 and d['config']['env_steps_per_bench_step'] == 3 * 48 * 7 * 2
  assert abs(d['value'] - 3 * d['config']['env_steps_per_bench_step'] / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
  rf = d['roofline']
  assert rf['windows'] == 3 and rf['frac_min'] <= rf['frac'] <= rf['frac_max']
  assert full['value'] == d['value'] and len(full['roofline']['windows_ms']) == 3       # the detail lives in the full object (stderr + bench_full.json)
  assert 'torch.distributed.run' in r.stderr                     # it said what it started


def test_gpus_2_line_carries_a_cpu_baseline_measured_before_the_ranks_exist():
  """N > 1: the self-launching parent (no torch, no GPU state) runs the line's CPU legs alone on the host and hands them to rank 0 of the child; cpu_baseline =
  the FASTEST CPU implementation of the run, the C oracle and the product's host build both named"""
  global ARGS
  args = [x for x in ARGS if x != '--no-cpu'] + ['--cpu-seconds', '0.2']
  env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', *args], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
  d, full = _the_line(r)
  cb = d['cpu_baseline']
  assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['impl'] in ('oracle_c', 'host_build') and cb['sample']
  assert cb['value'] >= max(cb['oracle_c'], cb['host_build']) * (1 - 1e-5)
  assert full['cpu_baseline_oracle_c']['by_threads'] and full['config0_host_build']['batch_env_steps_per_s'] > 0


def test_gpus_1_does_not_launch_anything():
  r = _run(1)
  d, _ = _the_line(r)
  assert d['n_gpus'] == 1 and d['cpu_baseline'] is None          # (--no-cpu)
  assert 'torch.distributed.run' not in r.stderr


def test_child_failure_is_the_exit_code():
  r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', *ARGS[:-4], '--test-env-factory', 'tests.bench_stub_env:no_such_factory'],
                     capture_output=True, text=True, timeout=600, cwd=REPO,
                     env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
  assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
