#!/usr/bin/env python3
"""bench.py -- env-steps/s of the tabletop_manipulation hot path on MI355X (BASELINE.json configs[1]).

One bench "step" = one pass of the hot path over one batch of synthetic input = ONE call of earl_tabletop_eval_episodes: E = 28
evaluation episodes (each: reset() + T = 200 wrapped env steps, the reference's eval horizon) of N = 4096 sparse-reward envs per GPU,
EVERY episode with its own U(-1,1) actions ([E, T, N, 3], pre-generated, resident in HBM: 275 MB), in ONE kernel launch whose shape does
not depend on --steps.  Every step's obs / reward / done / success is written to HBM exactly as E x T step() calls would (1.24 GB per
launch).  value = env-steps of all ranks / max-over-ranks wall time.  The episodes of a launch are independent (each starts with
reset()), so a 4096-env batch (64 workgroups on 256 CUs) has four of them in flight; the like-for-like figure with ONE episode in
flight per env is reported next to it (config.strict, and the `sequential_episodes` key).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--horizon 200] [--sweep] [--no-cpu]
  N > 1: either under the launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...), or
         plain `python bench.py --gpus N`: without WORLD_SIZE in the environment main() starts that launcher itself as a child process (self_launch)

Prints ONE JSON line (rank 0).  Extra keys: "roofline" (fused rollout kernel: algorithmic bytes / HIP-event time
vs 8 TB/s), "cpu_baseline" (the C oracle on the host cores, bounded sample), "step_api" (the same workload
through per-step launches of the gym-style step()), and with --sweep "sweep" (throughput vs N).
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# CPUs this process may run on, recorded BEFORE anything is imported: with OMP_PROC_BIND set, libgomp's load-time initialisation (it comes in with torch)
# binds the main thread to one place and a later sched_getaffinity() reports 1-2 CPUs (the CPU sweeps then topped out at 2 threads)
AVAIL_CPUS = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
# the reference's own simulators: probed, never assumed
REFERENCE_SIMULATORS = ('mujoco', 'mujoco_py', 'pybullet', 'pybullet_envs', 'dm_control', 'metaworld', 'gym')

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_ENV_STEP_ROLLOUT = 66  # act 12 B in; obs 48 + reward 4 + done 1 + success 1 B out
BYTES_PER_ENV_STEP_STEP = 144    # + fp64 state in/out per launch: qpos 2x32, attached 2x1, goal_idx 4, steps 2x4
STATE_BYTES_PER_ENV_LAUNCH = 2 * (32 + 1 + 4) + 4   # rollout: state read+written once per launch


def parse(argv=None):
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=20)       # 20 launches of 28 evaluation episodes each (the command tools/profile_bench.sh profiles)
  p.add_argument('--warmup', type=int, default=5)
  p.add_argument('--envs', type=int, default=4096, help='envs per GPU')
  p.add_argument('--horizon', type=int, default=200)
  p.add_argument('--reward', default='sparse')
  p.add_argument('--sweep', action='store_true', help='also report throughput vs N (rank 0, single GPU)')
  p.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
  p.add_argument('--cpu-seconds', type=float, default=10.0)
  p.add_argument('--no-step-api', action='store_true')
  p.add_argument('--episodes-per-launch', type=int, default=28,
                 help='evaluation episodes one bench step = one kernel launch walks (earl_tabletop_eval_episodes); 1 = one episode per launch')
  p.add_argument('--action-sets', type=int, default=4,
                 help='distinct action tensors [E, T, N, 3] the launches read round-robin (4 x 275 MB: nothing of an earlier launch survives in the 256 MiB Infinity Cache)')
  p.add_argument('--no-single', action='store_true', help='skip the one-episode-per-launch comparison leg')
  p.add_argument('--no-sawyer', action='store_true', help='skip the sawyer_door / sawyer_peg (BASELINE configs[2]) legs of the default line')
  p.add_argument('--sawyer-cpu-seconds', type=float, default=2.0, help='seconds per repetition and thread count of the Sawyer CPU baselines')
  p.add_argument('--no-kitchen', action='store_true', help='skip the kitchen (BASELINE configs[3]) leg of the default line')
  p.add_argument('--no-minitaur', action='store_true', help='skip the minitaur (BASELINE configs[4]) leg of the default line')
  p.add_argument('--settle-launches', type=int, default=40,
                 help='untimed launches of the headline shape BEFORE the --warmup ones: the board reaches its sustained power state (the first ~30 launches of a '
                      'process run 5-15 %% faster), so that `value` and `roofline.frac` are the sustained figures')
  p.add_argument('--roofline-windows', type=int, default=5,
                 help='timed windows of --steps launches each the roofline figure is taken over (median, with min / max); the FIRST window is the one `value` is timed in')
  p.add_argument('--test-env-factory', default=None, help=argparse.SUPPRESS)   # internal (tests/test_bench_launch.py): 'module:callable' -> the tabletop env of this rank; the launch
  p.add_argument('--test-backend', default=None, help=argparse.SUPPRESS)       # path (self-launch -> torch.distributed.run -> main) then runs over gloo on host tensors.  Never a product path.
  p.add_argument('--cpu-child', default=None, help=argparse.SUPPRESS)        # internal: run one CPU baseline in this (fresh, torch-free) process and print its JSON
  p.add_argument('--cpu-child-args', default='{}', help=argparse.SUPPRESS)
  p.add_argument('--workload', default='tabletop', choices=['tabletop', 'sawyer_door', 'sawyer_peg', 'kitchen', 'minitaur'],
                 help='tabletop = BASELINE configs[1] (the default, the quoted metric); sawyer_door / sawyer_peg = configs[2] shape, N=8192 each (next rows)')
  return p.parse_args(argv)


def make_env(torch, n, horizon, reward, rank, device):
  import earl_benchmark_amd as eb
  loader = eb.EARLEnvs('tabletop_manipulation', reward_type=reward, num_envs=n, device=device, seed=0,
                       env_offset=rank * n, eval_horizon=horizon, scalar_api=False)
  _, env = loader.get_envs()
  return env


def synth_actions(torch, T, n, rank, device, episodes=1):
  """U(-1,1) actions of one bench step: [T, n, 3], or [E, T, n, 3] -- every evaluation episode of the launch has its OWN actions, so all
  12 B per env-step of action reads come from HBM (round 2 replayed one [T, n, 3] array in every episode: those reads were cache hits)"""
  g = torch.Generator(device=device)
  g.manual_seed(1234 + rank)
  lead = (T, n) if episodes == 1 else (episodes, T, n)
  return (torch.rand(*lead, 3, generator=g, device=device) * 2 - 1).contiguous()


def alloc_out(torch, T, n, device, episodes=1):
  # zeros, not empty: the buffers are written once here, at allocation (a launch that is the FIRST to write a region of device memory runs
  # measurably slower, tools/archive/repro20.py); every warm-up launch then writes every row again -- the launch shape does not depend on --steps
  lead = (T, n) if episodes == 1 else (episodes, T, n)
  return (torch.zeros(*lead, 12, dtype=torch.float32, device=device), torch.zeros(*lead, dtype=torch.float32, device=device),
          torch.zeros(*lead, dtype=torch.bool, device=device), torch.zeros(*lead, dtype=torch.bool, device=device))


class _Clock:
  """HIP events on the launch stream (torch's current stream) when the job runs on a GPU; on CPU tensors (the gloo test of this very
  sequence, tests/test_bench_sequence.py) there is nothing to synchronise and the wall clock stands in for the events"""

  def __init__(self, torch, device):
    self.torch, self.cuda = torch, str(device).startswith('cuda')
    if self.cuda:
      self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

  def sync(self):
    if self.cuda:
      self.torch.cuda.synchronize()

  def start(self):
    self.t0 = time.perf_counter()
    if self.cuda:
      self.e0.record()

  def stop(self):
    if self.cuda:
      self.e1.record()
    self.t1 = time.perf_counter()

  def elapsed_ms(self):
    return self.e0.elapsed_time(self.e1) if self.cuda else (self.t1 - self.t0) * 1e3


def time_rollouts(torch, dist, env, acts, out, steps, warmup, world, device='cuda', gather_rollout=False):
  """The timed region of the job: W warm-up bench steps; barrier + synchronize; K bench steps; the ONE collective of the job (evaluation
  result of the last episode -> every rank); synchronize + barrier; wall time, MAX over ranks.  A bench step is ONE kernel launch:
  acts [E, T, n, 3] -> env.rollout_episodes (E evaluation episodes, each = reset + T env steps of the whole batch, each with its own
  actions; out has a leading episode axis), acts [T, n, 3] -> env.rollout(reset_first=True) (one episode).  Every launch of the job has
  the same shape, warm-up included.
  -> (seconds, [ms per launch from one event pair around the region], gathered [N_global, 2], gathered rollout or None, launches)"""
  from earl_benchmark_amd import sharding
  # `acts` may be a LIST of action tensors of one shape: launch j reads acts[j % len(acts)].  bench.main() passes several whose total
  # size is several times the 256 MiB Infinity Cache, so no launch finds the actions of an earlier one in a cache: every action byte of
  # every launch crosses the HBM interface (one tensor re-read by every launch read at 6.3-6.5 TB/s "algorithmic", 5.2-6.2 for real)
  sets = list(acts) if isinstance(acts, (list, tuple)) else [acts]
  acts = sets[0]
  multi = acts.dim() == 4
  issued = [0]

  def run(k):                                          # k bench steps = k launches
    for _ in range(k):
      a = sets[issued[0] % len(sets)]
      issued[0] += 1
      if multi:
        env.rollout_episodes(a, out=out)               # E x (reset + T steps): ONE kernel launch
      else:
        env.rollout(a, out=out, reset_first=True)      # reset() of every env + T steps: ONE kernel launch
    return k
  run(warmup)
  # one HIP event pair around the whole timed region, recorded on torch's current stream == the launch stream.
  # (An event pair per launch costs ~3 us of queue time per event on this stack.)
  clk = _Clock(torch, device)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  t0 = time.perf_counter()
  clk.start()
  launches = run(steps)
  clk.stop()
  gathered, rollout = None, None
  if world > 1:                 # the one collective of the job: evaluation result of the last episode -> every rank
    last = tuple(t[-1] for t in out) if multi else out
    sizes = [acts.shape[-2]] * world
    gathered = sharding.gather_summary(sharding.rollout_summary(last[1], last[3]), sizes=sizes)   # [N_global, 2]
    if gather_rollout:          # the whole [T, N/W, D+2] trajectory buffer of every rank -> [T, N, D+2]
      rollout = sharding.gather_rollout(sharding.pack_rollout(*last), sizes=sizes)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  kern_ms = [clk.elapsed_ms() / max(1, launches)]   # average launch duration over the timed region (incl. the inter-launch gap)
  return dt, kern_ms, gathered, rollout, launches


def time_step_api(torch, env, acts, steps, warmup, graph=False):
  """the same workload through the gym-style API, one step() launch per env step: eager (a ctypes call per step), or the T launches
  captured once by env.make_step_graph(T) and replayed with one host call per episode (the action ring holds the episode's actions)."""
  T = acts.shape[0]
  if graph:
    g = env.unwrapped.make_step_graph(T)
    g.actions.copy_(acts)

    def episode():
      env.reset()
      g.replay()
  else:
    def episode():
      env.reset()
      for t in range(T):
        env.step(acts[t])
  for _ in range(max(1, warmup // 4)):
    episode()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps):
    episode()
  e1.record()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return dt, e0.elapsed_time(e1) * 1e-3


def time_policy_in_the_loop(torch, n, T, reward, rank, device, steps, hidden=64):
  """The reference's TRAIN-env loop with a policy in it: LifelongWrapper(PersistentStateWrapper(env)) stepped one step() at a time, the action of step t + 1
  computed from the observation of step t by a 2-layer MLP (12 -> hidden -> 3, tanh) whose kernels are captured between the steps; T steps per replay, goal
  switches (every 50 steps) drawn inside the captured loop (device-resident Philox base).  -> dict for the `policy_in_the_loop` key"""
  import earl_benchmark_amd as eb
  L = eb.EARLEnvs('tabletop_manipulation', reward_type=reward, num_envs=n, device=device, seed=4321 + rank, setup_as_lifelong_learning=True, goal_change_frequency=50)
  env = L.get_envs()
  gen = torch.Generator(device=device).manual_seed(7 + rank)
  w1 = torch.randn(12, hidden, generator=gen, device=device) * 0.3
  w2 = torch.randn(hidden, 3, generator=gen, device=device) * 0.3

  def policy(ob):
    return torch.tanh(torch.tanh(ob @ w1) @ w2)
  env.reset()
  g = env.unwrapped.make_step_graph(T, policy=policy)
  g.replay()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  sgpu = e0.elapsed_time(e1) * 1e-3
  switched = int((env.unwrapped.steps_since_goal_change < 50).all())
  return {'value': steps * n * T / dt, 'unit': 'env-steps/s', 'launches': steps * T, 'us_per_env_step_wall': dt / (steps * T) * 1e6, 'us_per_env_step_gpu': sgpu / (steps * T) * 1e6,
          'policy': f'2-layer MLP 12 -> {hidden} -> 3 (tanh), fp32, torch kernels captured between the steps', 'goal_change_frequency': 50, 'goal_switch_bookkeeping_ok': bool(switched),
          'note': 'closed loop: the lifelong train env (wrappers/lifelong_wrapper.py:30-44) stepped by a captured policy; one replay = ' + str(T) + ' env steps; what a learner '
                  'in the loop sees, next to the open-loop evaluation figure `value` and the policy-free `step_api`'}


def probe_reference_simulators():
  """is any of the simulators the reference binds (env.yml:11-15: mujoco-py -> MuJoCo 2.1, pybullet, metaworld; plus their modern names)
  importable on THIS host?  find_spec first (no side effects), then a real import of what was found, in a child process (mujoco_py compiles at import)."""
  found, errors = [], {}
  for name in REFERENCE_SIMULATORS:
    try:
      if importlib.util.find_spec(name) is None:
        continue
    except (ImportError, ValueError) as e:
      errors[name] = repr(e)
      continue
    r = subprocess.run([sys.executable, '-c', f'import {name}; print(getattr({name}, "__version__", "?"))'], capture_output=True, text=True, timeout=300)
    if r.returncode == 0:
      found.append({'module': name, 'version': r.stdout.strip().splitlines()[-1] if r.stdout.strip() else '?'})
    else:
      errors[name] = (r.stderr.strip().splitlines() or ['import failed'])[-1][:200]
  names = [f['module'] for f in found]
  return {'probed': list(REFERENCE_SIMULATORS), 'found': found, 'import_errors': errors,
          'mujoco': any(n in names for n in ('mujoco', 'mujoco_py')), 'pybullet': 'pybullet' in names,
          'note': ('none of the reference\'s simulators is importable on this host: the CPU baselines below run this build\'s C restatement of its own stepper, and the '
                   'dynamics stay unpinned against MuJoCo / PyBullet (DESIGN.md)') if not found else
                  ('found ' + ', '.join(names) + (': tools/pin_with_simulator.py steps the build\'s own model description in it next to the HIP stepper'
                                                   if any(n in names for n in ('mujoco', 'mujoco_py', 'pybullet')) else
                                                   ': none of them is a simulator tools/pin_with_simulator.py can step a model in (it needs mujoco, mujoco_py or pybullet)'))}


def simulator_sentence(sim, which):
  """the clause the CPU-baseline samples carry about the real simulator, derived from the probe (never a constant)"""
  if sim is None:
    return f'{which} was not probed in this process'
  have = sim.get('mujoco') if which == 'MuJoCo' else sim.get('pybullet')
  return (f'{which} IS importable on this host (see reference_simulator)' if have else
          f'{which} itself is not importable on this host (probed: {", ".join(sim["probed"])}; found: {[f["module"] for f in sim["found"]] or "none"})')


def run_cpu_baseline_child(kind, **kw):
  """One CPU baseline in a FRESH child process (never a re-exec of this one): the OpenMP placement variables live in the child's environment only -- in
  the GPU process they would make libgomp pin the main thread, and every thread created after it (HIP runtime, RCCL proxy), to one core --
  and the child sizes its thread sweep from the CPUs it may use before it imports anything.  -> the baseline's dict"""
  env = dict(os.environ)
  env.setdefault('OMP_PROC_BIND', 'spread')
  env.setdefault('OMP_PLACES', 'cores')
  r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', kind, '--cpu-child-args', json.dumps(kw)], env=env, capture_output=True, text=True,
                     timeout=1800)
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  if r.returncode != 0 or not lines:
    return {'value': None, 'unit': 'env-steps/s', 'cores': None, 'kind': 'port', 'sample': f'CPU baseline child failed (rc {r.returncode}): {r.stderr.strip()[-300:]}'}
  return json.loads(lines[-1])


def cpu_child_main(kind, kw):
  """body of the child: no torch unless an imported module brings it; the main thread's affinity is restored after the imports"""
  sim = kw.pop('simulators', None)
  global SIMULATORS
  SIMULATORS = sim
  fn = {'tabletop': cpu_baseline, 'tabletop_host': host_build_leg, 'sawyer_door': lambda **k: sawyer_cpu_baseline(task='sawyer_door', **k), 'sawyer_peg': lambda **k: sawyer_cpu_baseline(task='sawyer_peg', **k),
        'kitchen': kitchen_cpu_baseline, 'minitaur': minitaur_cpu_baseline}[kind]
  res = fn(**kw)
  print(json.dumps(res), flush=True)


SIMULATORS = None                # result of probe_reference_simulators(), set by main() / handed to the children
CPU_RESULTS = {}                 # kind -> CPU baseline dict, filled by collect_cpu_baselines() BEFORE this process touches the GPU


def collect_cpu_baselines(a, world=1):
  """Every CPU leg of this run, each in its own fresh child process, BEFORE torch is imported here: child processes are started while this process has
  no GPU state, and the GPU process itself never carries the OpenMP placement variables.  Rank 0 only; at N > 1 only the legs of the line's own workload
  (the side workloads' CPU figures are N = 1 content)."""
  sim = SIMULATORS
  jobs = []
  if a.workload == 'tabletop':
    jobs.append(('tabletop', dict(n=a.envs, T=a.horizon, reward=a.reward, seconds=a.cpu_seconds)))
    jobs.append(('tabletop_host', dict(n=a.envs, T=a.horizon, reward=a.reward, seconds=2.0)))
    if world == 1 and not a.no_sawyer:
      jobs += [(w, dict(T_sample=0, seconds=a.sawyer_cpu_seconds, n=8192, simulators=sim)) for w in ('sawyer_door', 'sawyer_peg')]
    if world == 1 and not a.no_kitchen:
      jobs.append(('kitchen', dict(seconds=2.0, simulators=sim)))
    if world == 1 and not a.no_minitaur:
      jobs.append(('minitaur', dict(seconds=2.0, simulators=sim)))
  elif a.workload in ('sawyer_door', 'sawyer_peg'):
    jobs.append((a.workload, dict(T_sample=0, seconds=max(2.0, a.cpu_seconds / 5), n=a.envs if a.envs != 4096 else 8192, simulators=sim)))
  else:
    jobs.append((a.workload, dict(seconds=5.0, simulators=sim)))
  for kind, kw in jobs:
    CPU_RESULTS[kind] = run_cpu_baseline_child(kind, **kw)


def host_cpu_info():
  """what bounds the CPU baselines on this host: CPUs this process may run on, the cgroup CPU quota (a 256-CPU box with a quota of 16 CPUs' worth of
  time explains a thread sweep that peaks at 16), the OpenMP placement in force"""
  info = {'affinity_cpus': len(AVAIL_CPUS), 'os_cpu_count': os.cpu_count(),
          'OMP_PROC_BIND': os.environ.get('OMP_PROC_BIND'), 'OMP_PLACES': os.environ.get('OMP_PLACES'), 'cgroup_cpu_quota_cpus': None}
  try:
    if os.path.exists('/sys/fs/cgroup/cpu.max'):                          # cgroup v2: "<quota> <period>" or "max <period>"
      q, per = open('/sys/fs/cgroup/cpu.max').read().split()
      info['cgroup_cpu_quota_cpus'] = None if q == 'max' else float(q) / float(per)
    elif os.path.exists('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):           # cgroup v1
      q, per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()), int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
      info['cgroup_cpu_quota_cpus'] = None if q < 0 else q / per
  except Exception as e:      # noqa: BLE001  (diagnostic only)
    info['cgroup_cpu_quota_cpus'] = f'unreadable: {e}'
  return info


def cpu_baseline(n, T, reward, seconds):
  """oracle/ (C restatement of the reference, OpenMP over envs) on the host cores: bounded sample of the same
  workload.  Thread counts 1, 8, 16, ... up to the CPUs this process may use are each timed briefly; the fastest is
  then run for `seconds` and reported (cores = threads actually used)."""
  import numpy as np
  from oracle import tabletop_oracle as orc
  o = orc.OracleTabletop(n, reward_type=reward, horizon=T, seed=0)
  rng = np.random.default_rng(1234)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  avail = len(AVAIL_CPUS)

  out = (np.zeros((T, n, 12), np.float32), np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8), np.zeros((T, n), np.uint8))

  def run(threads, budget):
    orc.set_threads(threads)
    o.reset(); o.rollout(acts, out=out)      # warm-up (thread pool, page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
      o.reset()
      o.rollout(acts, out=out)
      reps += 1
      dt = time.perf_counter() - t0
      if dt >= budget:
        return reps, dt

  trials = {}
  for th in sorted({1, 4, 8, 16, 32, 64, 128, avail}):
    if th <= avail:
      reps, dt = run(th, 0.5)
      trials[th] = reps * n * T / dt
  best = max(trials, key=trials.get)
  reps, dt = run(best, seconds)
  # reference-shaped scalar loop: ONE env object, one step() call per env step from Python (the reference itself
  # spends >= 49 us per step in its Python/numpy layer alone)
  orc.set_threads(1)
  o1 = orc.OracleTabletop(1, reward_type=reward, horizon=T, seed=0)
  o1.reset()
  a1 = acts[:, :1].copy()
  k, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < 1.0:
    o1.step(a1[k % T])
    k += 1
  scalar_rate = k / (time.perf_counter() - t0)
  return {'value': reps * n * T / dt, 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
          'sample': f'{reps} rollouts of {n} envs x {T} steps ({reps * n * T} env-steps, {dt:.1f} s) through oracle/tabletop_oracle.c, '
                    f'OpenMP static over envs, {best} threads (fastest of {sorted(trials)}; host exposes {avail} CPUs)',
          'single_thread': trials[1], 'by_threads': {str(k): v for k, v in trials.items()},
          'scalar_python_loop_1env': scalar_rate, 'host': host_cpu_info()}


def host_build_leg(n, T, reward, seconds):
  """BASELINE configs[0] ("tabletop_manipulation sparse reward, 1 env, CPU ... plumbing, no GPU") on the PRODUCT's own host build (csrc/libearl_host.so: the
  kernels' per-env functions compiled by g++, include/earl_tabletop.h `_cpu` entry points; NOT the oracle): (a) EARLEnvs(num_envs=1, device='cpu'), the
  reference-shaped scalar loop -- reset() + T step() calls per episode, numpy in / the gym 4-tuple out; (b) the same library on the bench's own n-env batch,
  reset + fused T-step rollout, all threads."""
  import numpy as np
  import torch
  import earl_benchmark_amd as eb
  from earl_benchmark_amd import _abi
  _, env = eb.EARLEnvs('tabletop_manipulation', reward_type=reward, num_envs=1, device='cpu', eval_horizon=T).get_envs()
  rng = np.random.default_rng(0)
  acts = rng.uniform(-1, 1, (T, 3)).astype(np.float32)
  episodes, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    env.reset()
    for t in range(T):
      ob, rw, dn, info = env.step(acts[t])
    episodes += 1
  dt1 = time.perf_counter() - t0
  assert dn is True and isinstance(rw, float) and ob.shape == (12,)
  lib = _abi.load_host()
  quota = host_cpu_info().get('cgroup_cpu_quota_cpus')
  threads = lib.set_threads(max(1, min(len(AVAIL_CPUS), int(quota) if isinstance(quota, float) and quota >= 1 else len(AVAIL_CPUS))))
  _, benv = eb.EARLEnvs('tabletop_manipulation', reward_type=reward, num_envs=n, device='cpu', eval_horizon=T, scalar_api=False).get_envs()
  a = torch.from_numpy(rng.uniform(-1, 1, (T, n, 3)).astype(np.float32))
  out = benv.unwrapped._new_out((T, n))[0]
  benv.rollout(a, out=out, reset_first=True)
  reps, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < seconds:
    benv.rollout(a, out=out, reset_first=True)
    reps += 1
  dtn = time.perf_counter() - t0
  return {'scalar_env_steps_per_s': episodes * T / dt1, 'scalar_sample': f'{episodes} episodes of reset() + {T} step() calls on EARLEnvs(num_envs=1, device="cpu") ({dt1:.1f} s)',
          'batch_env_steps_per_s': reps * n * T / dtn, 'threads': threads, 'batch_sample': f'{reps} x (reset + {T}-step rollout) of {n} envs, {threads} OpenMP threads ({dtn:.1f} s)',
          'unit': 'env-steps/s', 'library': 'earl_benchmark_amd/csrc/libearl_host.so (csrc/tabletop_device.h compiled for the host; include/earl_tabletop.h *_cpu)',
          'workload': f'BASELINE configs[0]: tabletop_manipulation {reward} reward, 1 env, CPU'}


def sawyer_profile(workload, n, T):
  """static content of profiles/traffic.json for the Sawyer rollout kernel (written by tools/summarize_sawyer.py from separate rocprofv3
  --pmc passes over this same command): HBM bytes per launch and the SQ issue-slot shares; None when the profile is of another shape"""
  tpath = os.path.join(REPO, 'profiles', 'traffic.json')
  if not os.path.exists(tpath) or (n, T) != ((8192, 300) if workload == 'sawyer_door' else (8192, 200)):
    return {}
  return json.load(open(tpath)).get(workload, {})


def pipe_roofline(prof, kernel, gpu_ms, hbm=None):
  """roofline object of the articulated-body kernels: they are bound by ONE pipe of the SIMD, the vector ALU, not by HBM (state lives in LDS
  and registers for the whole rollout) and not by MFMA (block-sparse 9-23 wide matrices).  achieved = share of a SIMD's cycles in which its VALU
  issues = resident waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (the per-wave shares add up because a SIMD issues at most one VALU
  instruction per cycle; LDS and scalar issue run beside it and are listed, not summed); peak 1.0; no clamp.  The counters are STATIC content of
  profiles/ (separate rocprofv3 --pmc passes over this same workload), labelled so.  `wait` = share of wave cycles parked on s_waitcnt;
  `lane_occupancy` = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) when that pass was collected."""
  issue = prof.get('issue') or {}
  wps = prof.get('waves_per_simd') or 1
  valu = None if 'valu' not in issue else wps * issue['valu']
  return {'bound': 'valu', 'achieved': valu, 'peak': 1.0, 'frac': valu,
          'unit': 'share of SIMD cycles issuing a VALU instruction (waves per SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)',
          'waves_per_simd': wps, 'per_wave': {'valu': issue.get('valu'), 'lds': issue.get('lds'), 'scalar': issue.get('scalar'),
                                              'wait_s_waitcnt': issue.get('wait_any'), 'issue_stall': issue.get('wait_inst'), 'any_issue': issue.get('issue_any')},
          'lane_occupancy': issue.get('lane_occupancy'),
          # share of the SIMD's fp64 lane-cycles doing work: VALU issue share x lanes active per VALU instruction (the figure to raise)
          'valu_x_lane_occupancy': None if (valu is None or issue.get('lane_occupancy') is None) else valu * issue['lane_occupancy'],
          'source': (prof.get('source', '') + ' (static: SQ counters collected by rocprofv3 --pmc in separate runs of this workload, not measured in this run)') if issue else None,
          'kernel': kernel, 'kernel_ms_mean': gpu_ms, 'hbm': hbm}


def sawyer_cpu_baseline(T_sample, seconds, task='sawyer_door', n=8192, reps=3):
  """The C restatement of the same stepper and env loop (oracle/physics_oracle.c, OpenMP over envs) on the host cores.  MuJoCo itself
  availability is probed (probe_reference_simulators); this port runs the same algorithm the kernel runs.  Method: FIXED batch (n envs, the bench's own
  size) at every thread count; per count a short probe sizes a run of >= `seconds` (T env steps of random actions from the reset
  state), which is repeated `reps` times and the fastest repetition kept; the best count is reported."""
  import numpy as np
  from oracle import physics_c
  del T_sample
  cm = physics_c.CModel(task)
  rng = np.random.default_rng(0)
  if task == 'sawyer_door':
    hand = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
    r = cm.run(np.zeros((1, 10)), np.zeros((1, 10)), hand, [1, 0, 1, 0], [-1, 1], nsub=2000)      # sim.reset() + _reset_hand (converged)
    q0, v0 = r['qpos'][0].copy(), r['qvel'][0].copy()
    q0[9], v0[9] = -np.pi / 3, 0.0
    cfg = physics_c.door_cfg(att_names=cm.att_names)
  else:
    hand = np.array([0, 0.6, 0.2])
    r = cm.run(cm.tables['qpos0'][None], np.zeros((1, 15)), hand, [1, 0, 1, 0], [-1, 1], nsub=2000)
    q0, v0 = r['qpos'][0].copy(), r['qvel'][0].copy()
    q0[9:12], v0[9:] = [0.1, 0.6, 0.02], 0.0
    cfg = physics_c.peg_cfg(att_names=cm.att_names)
  acts_all = rng.uniform(-1, 1, (64, n, 4)).astype(np.float32)

  def run(threads, T):
    physics_c.set_threads(threads)
    q, v, mp = np.tile(q0, (n, 1)), np.tile(v0, (n, 1)), np.tile(hand, (n, 1))
    goal, steps = np.zeros((n, 7)), np.zeros(n, np.int32)
    acts = acts_all[:T] if T <= len(acts_all) else np.concatenate([acts_all] * (T // len(acts_all) + 1))[:T]
    t0 = time.perf_counter()
    cm.sawyer_rollout(cfg, q, v, mp, goal, steps, acts)
    return time.perf_counter() - t0
  ncpu = len(AVAIL_CPUS)
  cands = sorted({1, min(16, ncpu), min(64, ncpu), min(128, ncpu), ncpu})
  sweep, detail = {}, {}
  for c in cands:
    run(c, 1)                                                     # thread pool, page faults
    probe = run(c, 2) / 2                                         # seconds per env step of the batch
    T = max(2, int(np.ceil(seconds / probe)))
    best = min(run(c, T) for _ in range(reps))
    sweep[c] = n * T / best
    detail[c] = {'T': T, 'best_s': round(best, 3)}
  best = max(sweep, key=sweep.get)
  return {'value': sweep[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
          'sample': f'{n} envs x {detail[best]["T"]} env steps (5 timesteps each, random actions from the reset state), fastest of {reps} repetitions of '
                    f'>= {seconds:g} s each, through the C restatement of the same stepper (oracle/physics_oracle.c, OpenMP static over envs); the same '
                    f'{n}-env batch at every thread count: {({k: round(v) for k, v in sweep.items()})}; ' + simulator_sentence(SIMULATORS, 'MuJoCo'),
          'single_core': sweep[1], 'by_threads': {str(k): v for k, v in sweep.items()}, 'runs': {str(k): v for k, v in detail.items()}, 'host': host_cpu_info()}


def run_sawyer(a, torch, dist, world, rank, device, workload, steps, warmup, n=8192, T=None, cpu_seconds=None):
  """BASELINE configs[2] shape: n envs per GPU, reset + one fused T-step rollout per bench step, barrier + synchronize on both sides, max
  over ranks.  The dynamics are this build's own stepper (own contact model, parity with MuJoCo unpinned) -- see DESIGN.md.
  -> result dict on rank 0 (None elsewhere)"""
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from earl_benchmark_amd import sharding
  peg = workload == 'sawyer_peg'
  T = T or (200 if peg else 300)                         # the reference's eval horizons (earl_benchmark/__init__.py:24-35)
  kw = sharding.shard_kwargs(n * world, rank, world) if world > 1 else {}
  env = PersistentStateWrapper((SawyerPeg if peg else SawyerDoor)(reward_type=a.reward, num_envs=n, seed=1234,
                                                                  env_offset=kw.get('env_offset', rank * n)), T)
  nv, nq = env.unwrapped.nv, env.unwrapped.nq
  g = torch.Generator(device=device).manual_seed(99 + rank)
  acts = (torch.rand(T, n, 4, generator=g, device=device) * 2 - 1).to(torch.float32)
  out = env.unwrapped._new_out((T,))
  stream = torch.cuda.current_stream()

  def step():
    env.reset()
    env.rollout(acts, out=out)
  for _ in range(warmup):
    step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record(stream)
  for _ in range(steps):
    step()
  e1.record(stream)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  gpu_ms = e0.elapsed_time(e1) / steps
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  assert bool(out['done'][-1].all()) and not bool(out['done'][:-1].any()) and bool(torch.isfinite(out['obs']).all())
  diverged = int(out['status'].sum())
  if rank != 0:
    return None
  # algorithmic HBM bytes per env step: action 16 in; obs 14 x 8 + reward 4 + done 1 + success 1 + status 1 out; the last-stable state row
  # (qpos, qvel, mocap) written after every env step; per launch the state read once and the last observation written once
  state_row = (nq + nv + 3) * 8
  bytes_per_env_step = 16 + 14 * 8 + 4 + 1 + 1 + 1 + state_row
  per_launch = n * (T * bytes_per_env_step + state_row + 7 * 8 + 4 + 14 * 8)
  achieved = per_launch / (gpu_ms * 1e-3) / 1e9
  prof = sawyer_profile(workload, n, T)
  roof = pipe_roofline(prof, 'sawyer_rollout_kernel', gpu_ms,
                       hbm={'achieved_GBs': achieved, 'frac_of_8TBs': achieved / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': per_launch,
                            'bytes_per_env_step': bytes_per_env_step, 'traffic': prof.get('hbm_bytes_per_launch'),
                            'traffic_source': (prof.get('source', '') + ' (static)') if prof else None})
  res = {'value': steps * n * T * world / dt, 'unit': 'env-steps/s', 'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3,
         'kernel_ms': gpu_ms, 'timesteps_per_s': steps * n * T * world * 5 / dt, 'diverged_env_steps_last_rollout': diverged,
         'config': {'workload': f'{workload} {a.reward} reward, {n} batched envs per MI355X, reset + fused {T}-step rollout '
                                f'(5 timesteps per env step) per bench step; own stepper incl. contacts, parity with MuJoCo unpinned',
                    'schedule': ('one launch; persistent waves take (group of 4 envs, 10-step slice) items from a queue, least-advanced group first (csrc/physics.hip sched_claim)'
                                 if (peg and n > 4096) else 'one launch; one env group per wave' + (' (eight waves per CU)' if n > 4096 else '')),
                    'envs_per_gpu': n, 'horizon': T, 'frame_skip': 5, 'env_steps_per_bench_step': n * T * world,
                    'friction_cone': 'elliptic (the scene\'s MJCF; DESIGN.md 16.10)', 'reset_state': 'recorded (DESIGN.md 16.9)',
                    'parallelism': f'env-range shard x{world}, no per-step collective'},
         'valu_frac': roof['frac'], 'roofline': roof,
         'cpu_baseline': CPU_RESULTS.get(workload) if cpu_seconds is not None else None}
  del env, acts, out
  torch.cuda.empty_cache()
  return res


def kitchen_cpu_baseline(seconds, n=2048, reps=2):
  """The C restatement of the same stepper (oracle/physics_oracle.c, OpenMP over envs) and of the reference's numpy glue (oracle/glue_oracle.c)
  on the host cores: the bench's own 2048-env batch at every thread count, whole env steps (40 timesteps + action / observation / reward
  glue) of random actions from the reset state, >= `seconds` per repetition, fastest of `reps`."""
  import numpy as np
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  from earl_benchmark_amd import tables
  from oracle import glue_oracle as go, physics_c
  cm = physics_c.CModel('kitchen')
  names = cm.att_names
  site_idx = [names.index(s) for s in ('knob1_site', 'knob2_site', 'knob3_site', 'knob4_site', 'light_site', 'slide_site', 'hinge_site2', 'microhandle_site')]
  g = np.load(os.path.join(REPO, 'tests', 'golden', 'kitchen_step.npz'))
  p = go.kitchen_params(g['kitchen_pos_bound'], g['kitchen_vel_bound'], g['kitchen_pos_noise_amp'])
  goal = np.tile(tables.goal_states('kitchen')[0], (n, 1))
  mq = np.tile(cm.tables['weld_mocap_quat'], (n, 1))
  rng = np.random.default_rng(0)

  def run(threads, steps):
    physics_c.set_threads(threads)
    go.set_threads(threads) if hasattr(go, 'set_threads') else None
    q, v, mp = np.tile(INIT_QPOS, (n, 1)), np.zeros((n, 23)), np.tile(MIDPOINT_POS, (n, 1)).astype(np.float64)
    last = q[:, :9].copy()
    t0 = time.perf_counter()
    for _ in range(steps):
      a = rng.uniform(-1, 1, (n, 9)).astype(np.float32).astype(np.float64)
      mp, ctrl9 = go.kitchen_action(p, a, mp, last)
      r = cm.run(q, v, mp, mq, np.ascontiguousarray(ctrl9[:, :2]), nsub=40)
      q, v = r['qpos'], r['qvel']
      obs = go.kitchen_obs(p, q, goal, rng.uniform(-1, 1, (n, 46)))
      go.kitchen_reward(obs, mp, np.ascontiguousarray(r['att'][:, site_idx]))
      last = obs[:, :9].copy()
    return time.perf_counter() - t0
  ncpu = len(AVAIL_CPUS)
  sweep, detail = {}, {}
  for c in sorted({1, min(16, ncpu), min(64, ncpu), min(128, ncpu), ncpu}):
    probe = run(c, 1)
    steps = max(1, int(np.ceil(seconds / probe)))
    best = min(run(c, steps) for _ in range(reps))
    sweep[c] = n * steps / best
    detail[c] = {'env_steps': steps, 'best_s': round(best, 3)}
  best = max(sweep, key=sweep.get)
  return {'value': sweep[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
          'sample': f'{n} envs x {detail[best]["env_steps"]} env steps (40 timesteps each + the glue), fastest of {reps} repetitions of >= {seconds:g} s, through the C '
                    f'restatement of the same stepper (oracle/physics_oracle.c, OpenMP static over envs) and of the reference\'s numpy glue; the same {n}-env batch at '
                    f'every thread count: {({k: round(v) for k, v in sweep.items()})}; ' + simulator_sentence(SIMULATORS, 'MuJoCo'),
          'single_core': sweep[1], 'by_threads': {str(k): v for k, v in sweep.items()}, 'runs': {str(k): v for k, v in detail.items()}, 'host': host_cpu_info()}


# STATIC figures (not measured in this run): one MI355X, reset + one fused launch of a SHARD of the strong-scaling batches, relative to the full batch -- copied from
# profiles/r06_shard_launches.txt (tools/kitchen_small_batch.py on an MI355X, round 6; kitchen at 256 envs: four waves per env; at 512: two; minitaur: the 4096-env
# launch takes the two-waves-per-SIMD kernel, 139 ms, the shards the one-wave kernel, 80 - 87 ms).  What `world` GPUs
# would deliver if every shard ran like this one (no collective on the data path; the job's one all-gather is 16 KB)
SHARD_PROFILE = 'profiles/r06_shard_launches.txt'
MEASURED_SHARD_TIME = {'kitchen': {1: 1.00, 2: 0.95, 4: 0.76, 8: 0.60}, 'minitaur': {1: 1.00, 2: 0.66, 4: 0.62, 8: 0.62}}


def predicted_scaling(workload, n_global, world):
  """What the kernels' own layout predicts for configs[3] / [4] on `world` GPUs, both ways, stated in the line so that nobody reads 'x8' into the strong-scaling job
  (VERDICT r03 item 5a, r05 item 7).  An env is a serial chain of T x frame_skip timesteps walked by ONE 32-lane group; a launch lasts as long as its slowest wave.
    strong: the config as BASELINE.json words it -- a FIXED global batch range-sharded over the GPUs.  Sharding shortens a launch while a GPU has more waves than wave
            slots (kitchen: 2048 envs fill exactly one round on ONE GPU; minitaur: 4096 envs = two rounds), and below that only by what the small-batch launch modes
            recover (one env per wave / per workgroup, and, kitchen, two or four waves per env) -- STATIC ratios measured on one GPU per shard size, not extrapolated.
    weak:   the regime the design scales in -- the config's batch PER GPU (2048 / 4096 envs each): every rank runs the one-GPU launch on its own env range, no
            data-path collective, one all-gather of the [N, 2] evaluation summary per job (16-32 KB per rank: microseconds over xGMI), so the prediction is `world` x
            the one-GPU figure.  Nothing here is a measured multi-GPU number."""
  per_round = 2048                                      # the one-wave kernels: 2 envs per wave, 4 waves per CU, 256 CUs (the minitaur's two-wave kernel: 4096)
  rounds_1 = -(-n_global // per_round)
  rounds_w = -(-(-(-n_global // world)) // per_round)
  rel = MEASURED_SHARD_TIME.get(workload, {}).get(world) if n_global == (2048 if workload == 'kitchen' else 4096) else None
  strong = {'envs_per_gpu': -(-n_global // world), 'launch_rounds_on_1_gpu': rounds_1, f'launch_rounds_on_{world}_gpus': rounds_w,
            'predicted_speedup_vs_1_gpu': (1.0 / rel) if rel else rounds_1 / rounds_w,
            'basis': f'static: one MI355X running one shard of this size ({SHARD_PROFILE}, round 6; not measured in this run)' if rel else 'launch rounds',
            'note': (f'{workload}: {n_global} envs = {rounds_1} round(s) of 2048 resident envs on one MI355X; on {world} GPU(s) a shard of {-(-n_global // world)} envs takes '
                     + (f'{rel:.2f} x the full-batch launch, so the job is predicted {1.0 / rel:.2f} x faster' if rel else f'{rounds_w} round(s): predicted {rounds_1 / rounds_w:.1f} x at best')
                     + '.  The chain of an env (T x frame_skip dependent timesteps on one 32-lane group) does not shorten with more GPUs; the scaling lever of these workloads is MORE envs.')}
  weak = {'envs_per_gpu': n_global, 'envs_global': n_global * world, 'predicted_speedup_vs_1_gpu': float(world),
          'basis': 'design: independent env ranges, no data-path collective, one [N, 2] all-gather per job; every rank runs the one-GPU launch of this line',
          'all_gather_bytes_per_rank': n_global * 2 * 4}
  return {'strong': strong, 'weak': weak, 'predicted_speedup_vs_1_gpu': strong['predicted_speedup_vs_1_gpu'], 'basis': strong['basis']}


def run_kitchen(a, torch, dist, world, rank, device, steps, warmup, n_global=2048, T=400, cpu_seconds=None, env_factory=None):
  """BASELINE configs[3]: kitchen, 2048 envs in total range-sharded over the GPUs (n_global / world per GPU: STRONG scaling), one bench step =
  reset + T = 400 env steps (the reference's eval horizon) of 40 timesteps each in ONE fused launch (earl_kitchen_rollout); the per-step surface
  (env.step(): one launch per env step) is timed beside it as `step_api`.
  Own stepper, reduced collision set, parity with MuJoCo unpinned (envs/kitchen.py).  -> result dict on rank 0"""
  from earl_benchmark_amd.envs.kitchen import Kitchen
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from earl_benchmark_amd import sharding
  kw = sharding.shard_kwargs(n_global, rank, world)
  n = kw['num_envs']
  env = PersistentStateWrapper((env_factory or Kitchen)(num_envs=n, seed=1234, env_offset=kw['env_offset']), T)     # (env_factory: tests/test_bench_sequence.py's CPU stand-in)
  sizes = [sharding.shard_kwargs(n_global, r, world)['num_envs'] for r in range(world)]
  g = torch.Generator(device=device).manual_seed(77 + rank)
  acts = (torch.rand(T, n, 9, generator=g, device=device) * 2 - 1).to(torch.float32)

  outbuf = {}

  def episode():                                       # reset + ONE fused launch of T env steps (earl_kitchen_rollout), like the Sawyer lines
    env.reset()
    res = env.unwrapped.rollout(acts, out=outbuf)
    return res['obs'][-1], res['reward'][-1], res['done'][-1], {'status': res['status'][-1]}

  def stepped(k):                                      # the closed-loop surface: one launch per env step
    env.reset()
    for t in range(k):
      env.step(acts[t])
  for _ in range(warmup):
    episode()
  clk = _Clock(torch, device)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  t0 = time.perf_counter()
  clk.start()
  for _ in range(steps):
    o, r, done, info = episode()
  clk.stop()
  gathered = None
  if world > 1:                 # the ONE collective of the job: per-env return and final success of the last evaluation episode -> every rank ([N_global, 2])
    gathered = sharding.gather_summary(sharding.rollout_summary(outbuf['reward'].to(torch.float32), outbuf['success']), sizes=sizes)
    assert gathered.shape == (n_global, 2)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  assert bool(done.all()) and bool(torch.isfinite(o).all())
  fails = int(env.unwrapped.fail_count.sum())
  clk.sync()
  t1 = time.perf_counter()
  if not a.no_step_api:                                # (profiling runs skip it: env.step() launches the same kernel with T = 1, which would mix into its statistics)
    stepped(T)
  clk.sync()
  dt_step = time.perf_counter() - t1
  if rank != 0:
    return None
  prof = {}
  tpath = os.path.join(REPO, 'profiles', 'traffic.json')
  if os.path.exists(tpath) and (n_global, T, world) == (2048, 400, 1):
    prof = json.load(open(tpath)).get('kitchen', {})
  # algorithmic HBM bytes per env step: action 9 x 4 in; observation 46 x 8 + reward 8 + done 1 + success 1 + status 1 out; the last-stable state row (qpos, qvel: 2 x 23 x 8)
  # written after every env step.  Not what bounds the kernel (a few GB/s): reported because BASELINE.json asks for the achieved HBM fraction of every workload.
  kb = 36 + 46 * 8 + 8 + 3 + 2 * 23 * 8
  kms = clk.elapsed_ms() / steps
  khbm = {'achieved_GBs': n * T * kb / (kms * 1e-3) / 1e9, 'frac_of_8TBs': n * T * kb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bytes_per_env_step': kb,
          'algorithmic_bytes_per_launch': n * T * kb, 'traffic': prof.get('hbm_bytes_per_launch'), 'traffic_source': (prof.get('source', '') + ' (static)') if prof.get('hbm_bytes_per_launch') else None}
  roof = pipe_roofline(prof, 'kitchen_rollout_kernel', kms, hbm=khbm)
  return {'value': steps * n_global * T / dt, 'unit': 'env-steps/s', 'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3,
          'valu_frac': roof['frac'], 'roofline': roof,
          'timesteps_per_s': steps * n_global * T * 40 / dt, 'gpu_ms_per_env_step': clk.elapsed_ms() / (steps * T), 'scaling': 'strong',
          'diverged_env_steps': fails,
          'step_api': None if a.no_step_api else {'value': n * T / dt_step, 'unit': 'env-steps/s (this rank)', 'ms_per_env_step': dt_step / T * 1e3,
                       'note': 'the same episode through env.step(): one launch per env step (the rollout kernel with T = 1); every launch lasts as long as '
                               'its slowest wave, which the fused rollout only pays once per episode'},
          'config': {'workload': f'kitchen dense reward, {n_global} envs range-sharded over {world} MI355X ({n} per GPU), reset + one fused launch of {T} env steps of 40 '
                                 'timesteps per bench step; own stepper (nv = 23, 32 lanes per env), reduced collision set, parity with MuJoCo unpinned',
                     'envs_global': n_global, 'envs_per_gpu': n, 'horizon': T, 'frame_skip': 40, 'launches_per_episode': 1,
                     'parallelism': f'env-range shard x{world} of a FIXED {n_global}-env batch (strong scaling), no per-step collective, one all-gather of the [N, 2] '
                                    'evaluation summary per job', 'predicted_scaling': predicted_scaling('kitchen', n_global, world)},
          'gathered_rows': None if gathered is None else int(gathered.shape[0]),
          'cpu_baseline': CPU_RESULTS.get('kitchen') if cpu_seconds is not None else None}


def minitaur_cpu_baseline(seconds, n=4096, reps=2):
  """The C restatement of the same stepper and env loop (oracle/physics_oracle.c: oracle_minitaur_rollout, OpenMP over envs) on the host cores: the
  bench's own 4096-env batch at every thread count, whole env steps (leg model, 5 x (motor model + timestep), observation, reward) of random actions
  from the reset state, >= `seconds` per repetition, fastest of `reps`."""
  import numpy as np
  from oracle import physics_c
  rng = np.random.default_rng(0)
  c = physics_c.CMinitaur(n, seed=1234)
  physics_c.set_threads(len(AVAIL_CPUS))
  c.reset()
  q0, v0 = c.qpos.copy(), c.qvel.copy()

  def run(threads, steps):
    physics_c.set_threads(threads)
    c.qpos[:], c.qvel[:] = q0, v0
    c.overheat[:] = 0; c.motor_enabled[:] = 1
    acts = rng.uniform(-1, 1, (steps, n, 8)).astype(np.float32)
    t0 = time.perf_counter()
    c.rollout(acts)
    return time.perf_counter() - t0
  ncpu = len(AVAIL_CPUS)
  sweep, detail = {}, {}
  for k in sorted({1, min(16, ncpu), min(64, ncpu), min(128, ncpu), ncpu}):
    probe = run(k, 1)
    steps = max(1, int(np.ceil(seconds / probe)))
    best = min(run(k, steps) for _ in range(reps))
    sweep[k] = n * steps / best
    detail[k] = {'env_steps': steps, 'best_s': round(best, 3)}
  best = max(sweep, key=sweep.get)
  return {'value': sweep[best], 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
          'sample': f'{n} envs x {detail[best]["env_steps"]} env steps (5 timesteps each + motor model / observation / reward), fastest of {reps} repetitions of >= {seconds:g} s, '
                    f'through the C restatement of the same stepper (oracle/physics_oracle.c, OpenMP static over envs); the same {n}-env batch at every thread count: '
                    f'{({k: round(v) for k, v in sweep.items()})}; ' + simulator_sentence(SIMULATORS, 'PyBullet'),
          'single_core': sweep[1], 'by_threads': {str(k): v for k, v in sweep.items()}, 'runs': {str(k): v for k, v in detail.items()}, 'host': host_cpu_info()}


def run_minitaur(a, torch, dist, world, rank, device, steps, warmup, n_global=4096, T=1000, cpu_seconds=None, env_factory=None):
  """BASELINE configs[4]: minitaur, 4096 envs in total range-sharded over the GPUs (STRONG scaling, like the kitchen line), one bench step = reset
  (incl. its 100 settle timesteps) + T = 1000 env steps (the reference's eval horizon) of 5 timesteps each in ONE fused launch (earl_minitaur_rollout).
  Own robot model and stepper: parity with the reference's PyBullet simulation is unpinned and model-less (envs/minitaur.py).  -> result dict on rank 0"""
  from earl_benchmark_amd.envs.minitaur import Minitaur
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from earl_benchmark_amd import sharding
  kw = sharding.shard_kwargs(n_global, rank, world)
  n = kw['num_envs']
  env = PersistentStateWrapper((env_factory or Minitaur)(num_envs=n, seed=1234, env_offset=kw['env_offset'], scalar_api=False), T)
  sizes = [sharding.shard_kwargs(n_global, r, world)['num_envs'] for r in range(world)]
  g = torch.Generator(device=device).manual_seed(55 + rank)
  acts = (torch.rand(T, n, 8, generator=g, device=device) * 2 - 1).to(torch.float32)
  out = env.unwrapped._new_out((T,))

  def episode():
    env.reset()
    return env.rollout(acts, out=out)
  for _ in range(warmup):
    episode()
  clk = _Clock(torch, device)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  t0 = time.perf_counter()
  clk.start()
  for _ in range(steps):
    res = episode()
  clk.stop()
  gathered = None
  if world > 1:                 # the ONE collective of the job (see run_kitchen)
    gathered = sharding.gather_summary(sharding.rollout_summary(res['reward'].to(torch.float32), res['success']), sizes=sizes)
    assert gathered.shape == (n_global, 2)
  clk.sync()
  if world > 1:
    dist.barrier()
  clk.sync()
  dt = time.perf_counter() - t0
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  assert bool(res['done'][-1].all()) and not bool(res['done'][:-1].any()) and bool(torch.isfinite(res['obs']).all())
  fails = int(env.unwrapped.fail_count.sum())
  if rank != 0:
    return None
  prof = {}
  tpath = os.path.join(REPO, 'profiles', 'traffic.json')
  if os.path.exists(tpath) and (n_global, T, world) == (4096, 1000, 1):
    prof = json.load(open(tpath)).get('minitaur', {})
  # algorithmic HBM bytes per env step: action 8 x 4 in; observation 32 x 8 + reward 8 + done 1 + success 1 + status 1 out; the state row (qpos 23 + qvel 22 doubles) and the
  # motors' row (8 x (8 + 4 + 1)) written after every env step
  mb = 32 + 32 * 8 + 8 + 3 + (23 + 22) * 8 + 8 * 13
  mms = clk.elapsed_ms() / steps
  mhbm = {'achieved_GBs': n * T * mb / (mms * 1e-3) / 1e9, 'frac_of_8TBs': n * T * mb / (mms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'bytes_per_env_step': mb,
          'algorithmic_bytes_per_launch': n * T * mb, 'traffic': prof.get('hbm_bytes_per_launch'), 'traffic_source': (prof.get('source', '') + ' (static)') if prof.get('hbm_bytes_per_launch') else None}
  roof = pipe_roofline(prof, 'minitaur_duo_kernel (two waves per SIMD)' if (n >= 2049 and world == 1 and prof.get('waves_per_simd') == 2) else 'minitaur_kernel', mms, hbm=mhbm)
  return {'value': steps * n_global * T / dt, 'unit': 'env-steps/s', 'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3,
          'valu_frac': roof['frac'], 'roofline': roof, 'timesteps_per_s': steps * n_global * T * 5 / dt, 'scaling': 'strong', 'diverged_env_steps': fails,
          'config': {'workload': f'minitaur dense reward, {n_global} envs range-sharded over {world} MI355X ({n} per GPU), reset + one fused launch of {T} env steps of 5 '
                                 'timesteps per bench step; own robot model (nv = 22, four loop closures) and tree-structured stepper (csrc/minitaur_stepper.h), parity with PyBullet unpinned and model-less',
                     'envs_global': n_global, 'envs_per_gpu': n, 'horizon': T, 'frame_skip': 5, 'launches_per_episode': 2,
                     'launch_form': 'by batch size: the two-waves-per-SIMD rollout kernel when it needs less time for the batch (16 envs per CU: 4096 envs are one round), else the one-wave kernel; same bits',
                     'parallelism': f'env-range shard x{world} of a FIXED {n_global}-env batch (strong scaling), no per-step collective, one all-gather of the [N, 2] '
                                    'evaluation summary per job', 'predicted_scaling': predicted_scaling('minitaur', n_global, world)},
          'gathered_rows': None if gathered is None else int(gathered.shape[0]),
          'cpu_baseline': CPU_RESULTS.get('minitaur') if cpu_seconds is not None else None}


LINE_LIMIT = 4096                # bytes of the ONE stdout line (a 20 kB line was not parsed by the driver); everything else -> bench_full.json + stderr
FULL_PATH = os.path.join(REPO, 'bench_full.json')
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')


def _sig(x, digits=6):
  """floats of the side keys to `digits` significant figures (value / ms_per_step themselves stay exact); containers recursively"""
  if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
    return x
  if isinstance(x, float):
    return float(f'{x:.{digits}g}')
  if isinstance(x, dict):
    return {k: _sig(v, digits) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return [_sig(v, digits) for v in x]
  return x


def best_cpu(cpu, host):
  """cpu_baseline of the tabletop line = the FASTEST CPU implementation measured in this run: the C oracle (oracle/tabletop_oracle.c) or the product's
  own host build (csrc/libearl_host.so, the kernels' per-env functions compiled by g++), both named with their figures."""
  if not cpu and not host:
    return None
  cands = []
  if cpu and cpu.get('value'):
    cands.append(('oracle_c', cpu['value'], cpu.get('cores'), cpu.get('sample', '')))
  if host and host.get('batch_env_steps_per_s'):
    cands.append(('host_build', host['batch_env_steps_per_s'], host.get('threads'), host.get('batch_sample', '') + ' through csrc/libearl_host.so (include/earl_tabletop.h *_cpu)'))
  if not cands:
    return {'value': None, 'unit': 'env-steps/s', 'cores': None, 'kind': 'port', 'sample': (cpu or {}).get('sample', 'CPU legs failed')}
  name, v, cores, sample = max(cands, key=lambda c: c[1])
  return {'value': v, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port', 'impl': name, 'sample': sample[:300],
          'oracle_c': None if not cpu else cpu.get('value'), 'oracle_c_cores': None if not cpu else cpu.get('cores'),
          'host_build': None if not host else host.get('batch_env_steps_per_s'), 'host_build_cores': None if not host else host.get('threads'),
          'scalar_1env_python_loop': None if not host else host.get('scalar_env_steps_per_s')}


def compact_line(res):
  """The ONE stdout line: the contract's scalars, `config`, `roofline`, `cpu_baseline` -- each cut down to what the driver / judge reads (<= LINE_LIMIT bytes).
  `res` is the full result object (what rounds 1-5 printed); it goes to bench_full.json and to stderr untouched."""
  out = {k: res.get(k) for k in REQUIRED}
  c = res.get('config') or {}
  keep_c = ('workload', 'envs_per_gpu', 'global_envs', 'envs_global', 'horizon', 'frame_skip', 'episodes_per_bench_step', 'episodes_in_flight', 'env_instances_resident',
            'env_steps_per_bench_step', 'launches', 'parallelism')
  cfg = {k: c[k] for k in keep_c if k in c}
  if isinstance(cfg.get('workload'), str):
    cfg['workload'] = cfg['workload'][:200]
  if isinstance(cfg.get('parallelism'), str):
    cfg['parallelism'] = cfg['parallelism'][:120]
  st = c.get('strict')
  if st:
    cfg['strict'] = {'value': st.get('one_episode_in_flight'), 'frac': st.get('one_episode_in_flight_frac'), 'single_episode_launch': st.get('one_episode_per_launch'),
                     'single_episode_launch_frac': st.get('one_episode_per_launch_frac')}
  ow = c.get('other_workloads')
  if ow:
    cfg['other_workloads'] = {k: {kk: v.get(kk) for kk in ('value', 'ms', 'valu_x_lanes', 'cpu')} for k, v in ow.items()}
  ps = c.get('predicted_scaling')
  if ps:
    cfg['predicted_scaling'] = {'strong_speedup_vs_1_gpu': (ps.get('strong') or {}).get('predicted_speedup_vs_1_gpu'), 'strong_basis': (ps.get('strong') or {}).get('basis', '')[:110],
                                'weak_speedup_vs_1_gpu': (ps.get('weak') or {}).get('predicted_speedup_vs_1_gpu'), 'weak_envs_per_gpu': (ps.get('weak') or {}).get('envs_per_gpu')}
  out['config'] = cfg
  r = res.get('roofline') or {}
  keep_r = ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_min', 'frac_max', 'traffic', 'kernel', 'kernel_ms_median', 'kernel_ms_mean', 'strict_frac', 'strict_value',
            'bytes_per_env_step', 'algorithmic_bytes_per_launch', 'windows', 'waves_per_simd', 'lane_occupancy', 'valu_x_lane_occupancy')
  roof = {k: r[k] for k in keep_r if k in r}
  if isinstance(r.get('hbm'), dict):                        # the stepper lines: the VALU roofline is the bound; the HBM side rides along (BASELINE.json asks for it)
    roof['hbm'] = {k: r['hbm'].get(k) for k in ('achieved_GBs', 'frac_of_8TBs', 'bytes_per_env_step', 'traffic')}
  if isinstance(roof.get('unit'), str):
    roof['unit'] = roof['unit'][:80]
  cmpf = r.get('compare_with_profile')
  if cmpf:
    roof['compare_with_profile'] = {'frac': cmpf.get('frac'), 'kernel_avg_us': cmpf.get('kernel_avg_us'), 'profile': cmpf.get('profile')}
  out['roofline'] = roof
  cb = res.get('cpu_baseline')
  if cb:
    keep_b = ('value', 'unit', 'cores', 'kind', 'impl', 'sample', 'oracle_c', 'oracle_c_cores', 'host_build', 'host_build_cores', 'scalar_1env_python_loop', 'single_core')
    cb = {k: cb[k] for k in keep_b if k in cb}
    if isinstance(cb.get('sample'), str):
      cb['sample'] = cb['sample'][:300]
  out['cpu_baseline'] = cb
  out['full'] = os.path.basename(FULL_PATH)
  out = {k: (v if k in ('value', 'ms_per_step') else _sig(v)) for k, v in out.items()}
  line = json.dumps(out, separators=(',', ':'))
  if len(line) > LINE_LIMIT:                                  # never hand the driver a line it cannot parse: drop the optional blocks, longest first
    for victim in ('other_workloads', 'predicted_scaling'):
      out['config'].pop(victim, None)
      line = json.dumps(out, separators=(',', ':'))
      if len(line) <= LINE_LIMIT:
        break
  assert len(line) <= LINE_LIMIT, len(line)
  return line


def emit(res):
  """rank 0: the full object -> bench_full.json next to bench.py and -> stderr; the compact line -> stdout, LAST"""
  full = json.dumps(res)
  try:
    with open(FULL_PATH, 'w') as f:
      f.write(full + '\n')
  except OSError as e:
    print(f'bench.py: could not write {FULL_PATH}: {e}', file=sys.stderr)
  print('bench_full: ' + full, file=sys.stderr, flush=True)
  print(compact_line(res), flush=True)


def main_sawyer(a, torch, dist, world, rank, device):
  n = a.envs if a.envs != 4096 else 8192
  T = a.horizon if a.horizon != 200 else None
  r = run_sawyer(a, torch, dist, world, rank, device, a.workload, a.steps, a.warmup, n=n, T=T, cpu_seconds=None if a.no_cpu else max(2.0, a.cpu_seconds / 5))
  if rank == 0:
    res = {'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': r['value'], 'unit': 'env-steps/s', 'n_gpus': world, 'steps': a.steps,
           'warmup': a.warmup, 'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic', 'config': r['config'], 'roofline': r['roofline'], 'cpu_baseline': r['cpu_baseline'],
           'kernel_ms': r['kernel_ms'], 'valu_frac': r['valu_frac'], 'diverged_env_steps_last_rollout': r['diverged_env_steps_last_rollout'],
           'reference_simulator': SIMULATORS}
    emit(res)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


def self_launch(a, argv):
  """`python bench.py --gpus N` started WITHOUT a launcher (N > 1, no WORLD_SIZE in the environment): start the one-process-per-GPU job as a CHILD process
  -- `python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N bench.py <same arguments>` (the launcher picks its own
  rendezvous port: nothing to race for) --, relay its stdout (rank 0's ONE JSON line) and return its exit code.  A fresh child, never an exec of this
  process; this process has not imported torch nor touched a GPU, so it runs the CPU legs of the line FIRST, alone on the host cores, and hands them to rank 0 of the
  child through a file named in EARL_BENCH_CPU_RESULTS."""
  import tempfile
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  handover = None
  if not a.no_cpu:
    global SIMULATORS
    SIMULATORS = probe_reference_simulators()
    collect_cpu_baselines(a, world=a.gpus)
    fd, handover = tempfile.mkstemp(prefix='earl_bench_cpu_', suffix='.json')
    with os.fdopen(fd, 'w') as f:
      json.dump({'simulators': SIMULATORS, 'cpu': CPU_RESULTS}, f)
    env['EARL_BENCH_CPU_RESULTS'] = handover
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1', '--nproc-per-node', str(a.gpus),
         os.path.abspath(__file__), *argv]
  print(f'bench.py: --gpus {a.gpus} without a launcher: starting {" ".join(cmd[1:9])} ... as a child process', file=sys.stderr, flush=True)
  proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
  for line in proc.stdout:                                    # relay as it comes (stderr is inherited)
    sys.stdout.write(line)
    sys.stdout.flush()
  rc = proc.wait()
  if handover and os.path.exists(handover):
    os.unlink(handover)
  return rc


def main(argv=None):
  argv = list(sys.argv[1:] if argv is None else argv)
  a = parse(argv)
  if a.gpus > 1 and 'WORLD_SIZE' not in os.environ and not a.cpu_child:
    sys.exit(self_launch(a, argv))
  if a.cpu_child:                                             # a CPU baseline child of another bench.py: never touches the GPU
    cpu_child_main(a.cpu_child, json.loads(a.cpu_child_args))
    if hasattr(os, 'sched_setaffinity'):
      os.sched_setaffinity(0, AVAIL_CPUS)
    return
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC for RCCL: must be in the environment before the HIP runtime starts
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  # The reference's simulators: probed, and the CPU baselines: run -- each in a fresh child process that alone carries the OpenMP placement variables
  # (OMP_PROC_BIND / OMP_PLACES never enter THIS process: with them libgomp binds the main thread and everything created after it to one core) -- before
  # torch is imported here, i.e. before this process has any GPU state.
  global SIMULATORS
  if rank == 0:
    pre = os.environ.get('EARL_BENCH_CPU_RESULTS')             # a self-launching parent already ran the CPU legs (before any rank existed)
    if pre and os.path.exists(pre):
      h = json.load(open(pre))
      SIMULATORS = h['simulators']
      CPU_RESULTS.update(h['cpu'])
    else:
      SIMULATORS = probe_reference_simulators()
      if not a.no_cpu:                                         # N > 1 under a launcher: the line's own CPU legs only (the other ranks wait in the rendezvous meanwhile)
        collect_cpu_baselines(a, world)
  import torch
  import torch.distributed as dist
  if hasattr(os, 'sched_setaffinity') and 'OMP_PROC_BIND' not in os.environ:
    assert sorted(os.sched_getaffinity(0)) == AVAIL_CPUS, 'importing torch changed this process\'s CPU affinity'
  a.gpus = world                                             # (the launcher's WORLD_SIZE is what runs; main() self-launches when --gpus N > 1 came without one)
  stub = a.test_env_factory is not None                        # tests/test_bench_launch.py only: host tensors, gloo
  if stub:
    device = 'cpu'
    if world > 1:
      dist.init_process_group(a.test_backend or 'gloo')
  else:
    if not torch.cuda.is_available():
      sys.exit('bench.py needs an MI355X (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    if world > 1:
      dist.init_process_group('nccl', device_id=torch.device(device))
  if a.workload in ('sawyer_door', 'sawyer_peg'):
    return main_sawyer(a, torch, dist, world, rank, device)
  if a.workload == 'minitaur':
    r = run_minitaur(a, torch, dist, world, rank, device, min(a.steps, 3), min(a.warmup, 1), cpu_seconds=None if a.no_cpu else 5.0)
    if rank == 0:
      emit({'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': r['value'], 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': r['steps'], 'warmup': r['warmup'], 'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'config': r['config'], 'roofline': r['roofline'], 'cpu_baseline': r['cpu_baseline'],
            'timesteps_per_s': r['timesteps_per_s'], 'diverged_env_steps': r['diverged_env_steps'], 'reference_simulator': SIMULATORS})
    if world > 1:
      dist.barrier()
      dist.destroy_process_group()
    return
  if a.workload == 'kitchen':
    r = run_kitchen(a, torch, dist, world, rank, device, min(a.steps, 3), min(a.warmup, 1), cpu_seconds=None if a.no_cpu else 5.0)
    if rank == 0:
      emit({'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': r['value'], 'unit': 'env-steps/s', 'n_gpus': world,
            'steps': r['steps'], 'warmup': r['warmup'], 'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'config': r['config'], 'roofline': r['roofline'], 'cpu_baseline': r['cpu_baseline'],
            'timesteps_per_s': r['timesteps_per_s'], 'gpu_ms_per_env_step': r['gpu_ms_per_env_step'], 'diverged_env_steps': r['diverged_env_steps'],
            'reference_simulator': SIMULATORS})
    if world > 1:
      dist.barrier()
      dist.destroy_process_group()
    return
  n, T = a.envs, a.horizon
  E = max(1, a.episodes_per_launch)
  wgs = (n + 63) // 64
  in_flight = min(E, max(1, 256 // wgs)) if (E > 1 and wgs * 2 <= 256) else 1     # csrc/tabletop.hip do_rollout: episode groups side by side

  if a.test_env_factory:
    mod, fn = a.test_env_factory.split(':')
    env = getattr(importlib.import_module(mod), fn)(n=n, T=T, rank=rank, world=world)
  else:
    env = make_env(torch, n, T, a.reward, rank, device)
  R = max(1, a.action_sets)
  # [E, T, n, 3] x R: every episode of a launch has its own actions, and consecutive launches read different tensors
  act_sets = [synth_actions(torch, T, n, rank + 1000 * r, device, E) for r in range(R)]
  acts = act_sets[0]
  acts1 = acts[0] if E > 1 else acts
  out = alloc_out(torch, T, n, device, E)
  # the process's first launches (what rounds 1-3 quoted as `value`): 5-15 % faster than the sustained power state; kept as a side key
  edt, ekm, _, _, el = time_rollouts(torch, dist, env, act_sets, out, a.steps, a.warmup, world, device)
  early = {'value': a.steps * E * n * T * world / edt, 'unit': 'env-steps/s', 'kernel_ms_mean': ekm[0], 'launches': el,
           'note': f'the first launches of the process ({a.warmup} warm-up + {a.steps} timed), before the board settles into its sustained power state: NOT the headline'}
  # headline: --settle-launches more untimed launches first (about 10 ms of back-to-back traffic), then the --warmup ones, then EXACTLY --steps timed
  dt, kern_ms, gathered, _, launches = time_rollouts(torch, dist, env, act_sets, out, a.steps, a.warmup + max(0, a.settle_launches), world, device)
  if world > 1:
    assert gathered.shape == (n * world, 2)
  # roofline: the kernel's launch duration over several timed windows of the same shape (the first is the headline's own), median with min / max: one window
  # moved 0.64 <-> 0.71 of peak between boxes / runs in round 4
  window_ms = [kern_ms[0]]
  for _ in range(max(0, a.roofline_windows - 1)):
    _, wkm, _, _, _ = time_rollouts(torch, dist, env, act_sets, out, a.steps, 0, world, device)
    window_ms.append(wkm[0])
  dn = out[2] if E > 1 else out[2][None]
  assert bool(dn[:, -1].all()) and not bool(dn[:, :-1].any())        # done fires exactly at the horizon, in every episode
  if E > 1:
    assert not torch.equal(out[0][0], out[0][-1])                     # distinct actions -> distinct episodes
  single = None
  if E > 1 and not a.no_single:      # the same episodes, ONE per launch (round 1's bench step): the latency-bound regime of a 4096-env batch
    o1 = alloc_out(torch, T, n, device, 1)
    ks = max(8, a.steps * 2)
    sdt, skm, _, _, sl = time_rollouts(torch, dist, env, [x[e] for x in act_sets for e in range(E)] if E > 1 else act_sets, o1, ks, 4, world, device)
    single = {'value': ks * n * T * world / sdt, 'unit': 'env-steps/s', 'kernel_ms_mean': skm[0], 'launches': sl,
              'frac_of_8TBs': n * (T * BYTES_PER_ENV_STEP_ROLLOUT + STATE_BYTES_PER_ENV_LAUNCH) / (skm[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
              'note': 'one evaluation episode per launch (earl_tabletop_reset_rollout): what round 1 timed'}
    del o1
  sequential = None
  if E > 1 and not a.no_single and in_flight > 1:      # the same multi-episode launches with the episodes ONE AFTER THE OTHER (debug switch 38)
    from earl_benchmark_amd import _abi as _abi_
    lib_ = _abi_.load()
    lib_.earl_debug_set_rollout_impl(38)
    try:
      ks = max(2, a.steps // 2)
      qdt, qkm, _, _, ql = time_rollouts(torch, dist, env, act_sets, out, ks, 1, world, device)
    finally:
      lib_.earl_debug_set_rollout_impl(0)
    sequential = {'value': ks * E * n * T * world / qdt, 'unit': 'env-steps/s', 'kernel_ms_mean': qkm[0], 'launches': ql,
                  'frac_of_8TBs': n * (E * T * BYTES_PER_ENV_STEP_ROLLOUT + STATE_BYTES_PER_ENV_LAUNCH) / (qkm[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  'note': f'the same launches ({E} evaluation episodes each, own actions) with ONE episode in flight per env: the episodes of a launch one '
                          'after the other on the batch\'s 64 workgroups -- the reference\'s evaluation loop taken literally'}
  total_env_steps = a.steps * E * n * T * world
  value = total_env_steps / dt
  # cross-check of the headline: 60 more of the same launches, the last 40 timed (tools/archive/placement_experiment.py: the first ~30 launches of a process run
  # 5-15 % faster than the ones after ~10 ms of back-to-back traffic); `value` itself is measured after --settle-launches + --warmup untimed launches
  sustained = None
  if E > 1 and not a.no_single:
    sdt2, skm2, _, _, _ = time_rollouts(torch, dist, env, act_sets, out, 40, 20, world, device)
    sustained = {'value': 40 * E * n * T * world / sdt2, 'unit': 'env-steps/s', 'kernel_ms_mean': skm2[0],
                 'frac_of_8TBs': n * (E * T * BYTES_PER_ENV_STEP_ROLLOUT + STATE_BYTES_PER_ENV_LAUNCH) / (skm2[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 'note': 'cross-check of `value`: 40 more of the same launches timed after 20 more untimed ones'}

  # BASELINE configs[2] in the same run (every rank takes part: same barrier / max-over-ranks timing; the CPU legs at N = 1 only)
  sawyer = {}
  if not a.no_sawyer:
    for w in ('sawyer_door', 'sawyer_peg'):
      sawyer[w] = run_sawyer(a, torch, dist, world, rank, device, w, steps=3, warmup=2,
                             cpu_seconds=None if (a.no_cpu or world > 1) else a.sawyer_cpu_seconds)
  if not a.no_kitchen:      # BASELINE configs[3] in the same run (2048 envs in total, sharded over the ranks)
    sawyer['kitchen'] = run_kitchen(a, torch, dist, world, rank, device, steps=2, warmup=1, cpu_seconds=None if (a.no_cpu or world > 1) else 2.0)
  if not a.no_minitaur:     # BASELINE configs[4] in the same run (4096 envs in total, sharded over the ranks)
    sawyer['minitaur'] = run_minitaur(a, torch, dist, world, rank, device, steps=2, warmup=1, cpu_seconds=None if (a.no_cpu or world > 1) else 2.0)
  res = None
  if rank == 0:
    kmean = sorted(window_ms)[len(window_ms) // 2]             # median launch duration over the timed windows
    # algorithmic bytes of a launch = bytes that cross the HBM interface: its E episodes x T steps x (12 B of actions in -- each episode
    # has its own --, obs 48 + reward 4 + done 1 + success 1 out) + the env state once (66 B per env-step fused)
    bytes_per_launch = n * (E * T * BYTES_PER_ENV_STEP_ROLLOUT + STATE_BYTES_PER_ENV_LAUNCH)
    achieved = bytes_per_launch / (kmean * 1e-3) / 1e9
    traffic, traffic_source, compare = None, None, None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
      tj = json.load(open(tpath))
      key = f'eval_n{n}_T{T}_E{E}_own_actions' if E > 1 else f'rollout_n{n}_T{T}'
      traffic = tj.get(key, {}).get('hbm_bytes_per_launch')
      pns = tj.get(key, {}).get('rocprof_kernel_average_ns')
      if pns:                                                   # the committed rocprofv3 --kernel-trace --stats figure this line's `frac` is to be compared with
        compare = {'profile': tj[key].get('stats', tj[key].get('source')), 'kernel_avg_us': pns / 1e3,
                   'frac': bytes_per_launch / (pns * 1e-9) / 1e9 / HBM_PEAK_GBS,
                   'note': 'STATIC: average launch duration of this kernel in the committed rocprofv3 --kernel-trace --stats run of this same command (another box, another '
                           'day); this run\'s live figure is `frac` (median of `windows_ms`, min / max beside it)'}
      if traffic is not None:
        traffic_source = f"profiles/traffic.json <- {tj[key].get('source')} (static: FETCH_SIZE / WRITE_SIZE passes of rocprofv3 --pmc over this command -- same launch shape --, not measured in this run)"
    strict = {'one_episode_in_flight': None if sequential is None else sequential['value'],
              'one_episode_in_flight_frac': None if sequential is None else sequential['frac_of_8TBs'],
              'one_episode_per_launch': None if single is None else single['value'], 'one_episode_per_launch_frac': None if single is None else single['frac_of_8TBs'],
              'sustained': None if sustained is None else sustained['value'], 'sustained_frac': None if sustained is None else sustained['frac_of_8TBs']}
    res = {
        'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'tabletop_manipulation {a.reward}, {n} envs/GPU; bench step = 1 launch = {E} eval episodes (reset+{T} steps), own actions',
                   'envs_per_gpu': n, 'episodes_per_bench_step': E, 'episodes_in_flight': in_flight, 'env_instances_resident': n * in_flight,
                   'strict': strict, 'launches': launches, 'settle_launches': max(0, a.settle_launches), 'untimed_launches_before_the_timed_region': a.warmup + a.steps + a.warmup + max(0, a.settle_launches),
                   'early_window': early['value'], 'global_envs': n * world, 'horizon': T, 'env_steps_per_bench_step': E * n * T * world,
                   'actions': f'{R} x [{E}, {T}, {n}, 3] f32 per GPU ({R * E * T * n * 12 / 1e6:.0f} MB): distinct per episode, launches read the {R} tensors round-robin',
                   'action_sets': R,
                   'next_rows': {k: (v or {}).get('value') for k, v in sawyer.items()},
                   # BASELINE configs[2..4] timed in this same run, compact (the full objects are the top-level keys of the same names)
                   'other_workloads': {k: {'value': v.get('value'), 'unit': 'env-steps/s', 'ms': v.get('ms_per_step'), 'envs': v['config'].get('envs_per_gpu'),
                                           'horizon': v['config'].get('horizon'), 'valu_x_lanes': (v.get('roofline') or {}).get('valu_x_lane_occupancy'),
                                           'cpu': ((v.get('cpu_baseline') or {}).get('value')), 'cpu_cores': ((v.get('cpu_baseline') or {}).get('cores'))}
                                       for k, v in sawyer.items() if v},
                   'parallelism': f'env-range shard x{world}, no per-step collective'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source, 'kernel': 'rollout_ws_kernel',
                     'kernel_ms_mean': sum(window_ms) / len(window_ms), 'kernel_ms_median': kmean, 'windows_ms': window_ms, 'windows': len(window_ms), 'launches_per_window': a.steps,
                     'frac_min': bytes_per_launch / (max(window_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 'frac_max': bytes_per_launch / (min(window_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'frac_headline_window': bytes_per_launch / (window_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 'compare_with_profile': compare,
                     'algorithmic_bytes_per_launch': bytes_per_launch,
                     'bytes_per_env_step': BYTES_PER_ENV_STEP_ROLLOUT, 'episodes_per_launch': E, 'launches': launches,
                     'strict_frac': strict['one_episode_in_flight_frac'], 'strict_value': strict['one_episode_in_flight']},
    }
    res['early_window'] = early
    res['reference_simulator'] = SIMULATORS
    if single is not None:
      res['single_episode_launch'] = single
    if sequential is not None:
      res['sequential_episodes'] = sequential
    if sustained is not None:
      res['sustained'] = sustained
    if not a.no_step_api:
      env2 = make_env(torch, n, T, a.reward, rank, device)
      ks = max(1, a.steps // 10)

      def leg(graph):
        sdt, sgpu = time_step_api(torch, env2, acts1, ks, a.warmup, graph=graph)
        return {'value': ks * n * T / sdt, 'unit': 'env-steps/s', 'launches': ks * T, 'us_per_step_call_wall': sdt / (ks * T) * 1e6,
                'us_per_step_call_gpu': sgpu / (ks * T) * 1e6, 'achieved_GBs': n * BYTES_PER_ENV_STEP_STEP / (sgpu / (ks * T)) / 1e9}
      res['step_api'] = leg(True)
      res['step_api']['note'] = ('same workload, one step() kernel launch per env step, the T launches of an episode captured by '
                                 'env.make_step_graph(T) and replayed with one host call (closed-loop users capture their policy in between)')
      res['step_api']['eager'] = leg(False)
      res['step_api']['eager']['note'] = 'one eager step() call per env step from Python (host-launch bound at N=4096)'
      res['policy_in_the_loop'] = time_policy_in_the_loop(torch, n, T, a.reward, rank, device, max(2, a.steps // 4))
    if a.sweep:
      sw = []
      for ns in (64, 1024, 4096, 16384, 65536, 262144, 1048576):
        Ts = T if ns * T * 66 < 12e9 else max(8, int(12e9 // (ns * 66)))
        e = make_env(torch, ns, Ts, a.reward, 0, device)
        ac = synth_actions(torch, Ts, ns, 0, device)
        o = alloc_out(torch, Ts, ns, device)
        k = max(3, min(a.steps, int(2e9 // (ns * Ts * 66)) + 3))
        d, km, _, _, _ = time_rollouts(torch, dist, e, ac, o, k, 3, 1, device)
        kmn = sum(km) / len(km)
        sw.append({'envs': ns, 'T': Ts, 'env_steps_per_s': k * ns * Ts / d, 'kernel_ms': kmn,
                   'GBs': ns * (Ts * 66 + STATE_BYTES_PER_ENV_LAUNCH) / (kmn * 1e-3) / 1e9})
        del e, ac, o
        torch.cuda.empty_cache()
      res['sweep'] = sw
    if not a.no_cpu:
      res['cpu_baseline'] = best_cpu(CPU_RESULTS.get('tabletop'), CPU_RESULTS.get('tabletop_host'))     # the fastest CPU implementation of this run, both named
      res['cpu_baseline_oracle_c'] = CPU_RESULTS.get('tabletop')
      res['config0_host_build'] = CPU_RESULTS.get('tabletop_host')          # BASELINE configs[0] on the product's own `_cpu` entry points
      hb = CPU_RESULTS.get('tabletop_host') or {}
      res['config']['config0_cpu_1env'] = {'scalar_env_steps_per_s': hb.get('scalar_env_steps_per_s'), 'batch_env_steps_per_s': hb.get('batch_env_steps_per_s')}
    else:
      res['cpu_baseline'] = None
    res.update(sawyer)            # "sawyer_door": {...}, "sawyer_peg": {...}: value, kernel_ms, issue_frac, roofline, cpu_baseline
    emit(res)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
