#!/usr/bin/env python3
"""bench.py -- env-steps/s of the tabletop_manipulation hot path on MI355X (BASELINE.json configs[1]).

One bench "step" = one evaluation rollout of the whole batch: reset() + T = 200 wrapped env steps (the reference's
eval horizon) of N = 4096 sparse-reward envs per GPU, executed by ONE fused reset+rollout kernel launch.
Actions are synthetic U(-1,1) (pre-generated, resident in HBM); every step's obs / reward / done / success is
written to HBM exactly as T step() calls would.  value = env-steps of all ranks / max-over-ranks wall time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--horizon 200] [--sweep] [--no-cpu]
  N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra keys: "roofline" (fused rollout kernel: algorithmic bytes / HIP-event time
vs 8 TB/s), "cpu_baseline" (the C oracle on the host cores, bounded sample), "step_api" (the same workload
through per-step launches of the gym-style step()), and with --sweep "sweep" (throughput vs N).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BYTES_PER_ENV_STEP_ROLLOUT = 66  # act 12 B in; obs 48 + reward 4 + done 1 + success 1 B out
BYTES_PER_ENV_STEP_STEP = 144    # + fp64 state in/out per launch: qpos 2x32, attached 2x1, goal_idx 4, steps 2x4
STATE_BYTES_PER_ENV_LAUNCH = 2 * (32 + 1 + 4) + 4   # rollout: state read+written once per launch


def parse():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=200)
  p.add_argument('--warmup', type=int, default=20)
  p.add_argument('--envs', type=int, default=4096, help='envs per GPU')
  p.add_argument('--horizon', type=int, default=200)
  p.add_argument('--reward', default='sparse')
  p.add_argument('--sweep', action='store_true', help='also report throughput vs N (rank 0, single GPU)')
  p.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
  p.add_argument('--cpu-seconds', type=float, default=10.0)
  p.add_argument('--no-step-api', action='store_true')
  p.add_argument('--workload', default='tabletop', choices=['tabletop', 'sawyer_door', 'sawyer_peg'],
                 help='tabletop = BASELINE configs[1] (the default, the quoted metric); sawyer_door / sawyer_peg = configs[2] shape, N=8192 each (next rows)')
  return p.parse_args()


def make_env(torch, n, horizon, reward, rank, device):
  import earl_benchmark_amd as eb
  loader = eb.EARLEnvs('tabletop_manipulation', reward_type=reward, num_envs=n, device=device, seed=0,
                       env_offset=rank * n, eval_horizon=horizon, scalar_api=False)
  _, env = loader.get_envs()
  return env


def synth_actions(torch, T, n, rank, device):
  g = torch.Generator(device=device)
  g.manual_seed(1234 + rank)
  return (torch.rand(T, n, 3, generator=g, device=device) * 2 - 1).contiguous()


def alloc_out(torch, T, n, device):
  return (torch.empty(T, n, 12, dtype=torch.float32, device=device), torch.empty(T, n, dtype=torch.float32, device=device),
          torch.empty(T, n, dtype=torch.bool, device=device), torch.empty(T, n, dtype=torch.bool, device=device))


def time_rollouts(torch, dist, env, acts, out, steps, warmup, world):
  """K x (reset + T steps, one launch each), barrier + synchronize on both sides; HIP events around every launch."""
  for _ in range(warmup):
    env.rollout(acts, out=out, reset_first=True)
  # one HIP event pair around the whole timed region, recorded on torch's current stream == the launch stream.
  # (An event pair per launch costs ~3 us of queue time per event on this stack -- 20 % of a 34 us kernel.)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0.record()
  for k in range(steps):
    env.rollout(acts, out=out, reset_first=True)     # reset() of every env + T steps: ONE kernel launch
  e1.record()
  gathered = None
  if world > 1:                 # the one collective of the job: evaluation result of the last rollout -> every rank
    from earl_benchmark_amd import sharding
    gathered = sharding.gather_summary(sharding.rollout_summary(out[1], out[3]), sizes=[acts.shape[1]] * world)   # [N_global, 2]
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  kern_ms = [e0.elapsed_time(e1) / steps]   # average launch duration over the timed region (incl. the inter-launch gap)
  return dt, kern_ms, gathered


def time_step_api(torch, env, acts, steps, warmup):
  """the same workload through the gym-style API: one step() launch per env step (eager, no graph)."""
  T = acts.shape[0]
  for _ in range(max(1, warmup // 4)):
    env.reset()
    for t in range(T):
      env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps):
    env.reset()
    for t in range(T):
      env.step(acts[t])
  e1.record()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return dt, e0.elapsed_time(e1) * 1e-3


def cpu_baseline(n, T, reward, seconds):
  """oracle/ (C restatement of the reference, OpenMP over envs) on the host cores: bounded sample of the same
  workload.  Thread counts 1, 8, 16, ... up to the CPUs this process may use are each timed briefly; the fastest is
  then run for `seconds` and reported (cores = threads actually used)."""
  import numpy as np
  from oracle import tabletop_oracle as orc
  o = orc.OracleTabletop(n, reward_type=reward, horizon=T, seed=0)
  rng = np.random.default_rng(1234)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)

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
          'scalar_python_loop_1env': scalar_rate}


def sawyer_traffic(workload, n, T):
  """HBM bytes per launch of the Sawyer rollout kernel from the committed PMC profile (profiles/traffic.json, written by
  tools/summarize_sawyer.py from separate FETCH_SIZE / WRITE_SIZE passes), or None when the profile is of another shape"""
  tpath = os.path.join(REPO, 'profiles', 'traffic.json')
  if not os.path.exists(tpath) or (n, T) != ((8192, 300) if workload == 'sawyer_door' else (8192, 200)):
    return None
  return json.load(open(tpath)).get(workload, {}).get('hbm_bytes_per_launch')


def sawyer_cpu_baseline(T_sample, seconds, task='sawyer_door'):
  """The C restatement of the same stepper and env loop (oracle/physics_oracle.c, OpenMP over envs) on the host cores:
  a short thread sweep, then a bounded sample at the best thread count.  MuJoCo itself is not available on this host;
  this port runs the same algorithm the kernel runs (3.8 us per timestep per core where it was written)."""
  import numpy as np
  from oracle import physics_c
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

  def run(threads, n, T):
    physics_c.set_threads(threads)
    q, v, mp = np.tile(q0, (n, 1)), np.tile(v0, (n, 1)), np.tile(hand, (n, 1))
    goal, steps = np.zeros((n, 7)), np.zeros(n, np.int32)
    acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
    t0 = time.perf_counter()
    cm.sawyer_rollout(cfg, q, v, mp, goal, steps, acts)
    return n * T / (time.perf_counter() - t0)
  ncpu = len(os.sched_getaffinity(0))
  cands = sorted({1, min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu), ncpu})
  sweep = {c: run(c, 64 * c, 10) for c in cands}
  best = max(sweep, key=sweep.get)
  n = 64 * best
  T = max(10, min(T_sample, int(seconds * sweep[best] / n)))
  val = run(best, n, T)
  return {'value': val, 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
          'sample': f'{n} envs x {T} env steps (5 timesteps each, random actions from the reset state) through the C restatement of the '
                    f'same stepper (oracle/physics_oracle.c, OpenMP); thread sweep {({k: round(v) for k, v in sweep.items()})}; '
                    'MuJoCo itself is not available on this host',
          'single_core': sweep[1]}


def main_sawyer(a, torch, dist, world, rank, device):
  """BASELINE configs[2] shape (Sawyer door half): N envs per GPU, reset + one fused T-step rollout per bench step.
  The dynamics are this build's own stepper (own contact model, parity with MuJoCo unpinned) -- see DESIGN.md."""
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from earl_benchmark_amd import sharding
  peg = a.workload == 'sawyer_peg'
  n = a.envs if a.envs != 4096 else 8192
  T = a.horizon if a.horizon != 200 else (200 if peg else 300)          # the reference's eval horizons (earl_benchmark/__init__.py:24-35)
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
  for _ in range(a.warmup):
    step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record(stream)
  for _ in range(a.steps):
    step()
  e1.record(stream)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  dt = time.perf_counter() - t0
  gpu_ms = e0.elapsed_time(e1) / a.steps
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  assert bool(out['done'][-1].all()) and not bool(out['done'][:-1].any()) and bool(torch.isfinite(out['obs']).all())
  if rank == 0:
    bytes_per_env_step = 16 + 14 * 8 + 4 + 1 + 1          # action + obs (f64) + reward + done + success
    per_launch = n * (T * bytes_per_env_step + 2 * ((nq + nv) * 8 + 3 * 8) + 7 * 8 + 4 * 2)
    achieved = per_launch / (gpu_ms * 1e-3) / 1e9
    res = {'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': a.steps * n * T * world / dt, 'unit': 'env-steps/s',
           'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'{a.workload} {a.reward} reward, {n} batched envs per MI355X, reset + fused {T}-step rollout '
                                  f'(5 timesteps per env step) per bench step; own stepper incl. contacts, parity with MuJoCo unpinned',
                      'envs_per_gpu': n, 'horizon': T, 'frame_skip': 5, 'env_steps_per_bench_step': n * T * world,
                      'parallelism': f'env-range shard x{world}, no per-step collective'},
           'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                        'traffic': sawyer_traffic(a.workload, n, T), 'kernel': 'sawyer_rollout_kernel', 'kernel_ms_mean': gpu_ms,
                        'algorithmic_bytes_per_launch': per_launch, 'bytes_per_env_step': bytes_per_env_step,
                        'note': 'not HBM-bound: instruction issue and LDS latency at one wave per SIMD (door: 55 % of wave cycles issue, '
                                '41 % wait on LDS / memory counters, profiles/r01_sawyer_door_rollout_pmc.json); the HBM figure is reported '
                                'because the schema asks for it'},
           'cpu_baseline': None if a.no_cpu else sawyer_cpu_baseline(T, a.cpu_seconds, a.workload)}
    print(json.dumps(res), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


def main():
  a = parse()
  import torch
  import torch.distributed as dist
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != a.gpus:
    if world == 1 and a.gpus > 1:
      sys.exit(f'--gpus {a.gpus} needs the torch.distributed.run launcher (one process per GPU)')
    a.gpus = world
  if not torch.cuda.is_available():
    sys.exit('bench.py needs an MI355X (the hot path has no CPU fallback)')
  torch.cuda.set_device(local_rank)
  device = f'cuda:{local_rank}'
  if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', device_id=torch.device(device))
  if a.workload in ('sawyer_door', 'sawyer_peg'):
    return main_sawyer(a, torch, dist, world, rank, device)
  n, T = a.envs, a.horizon

  env = make_env(torch, n, T, a.reward, rank, device)
  acts = synth_actions(torch, T, n, rank, device)
  out = alloc_out(torch, T, n, device)
  dt, kern_ms, gathered = time_rollouts(torch, dist, env, acts, out, a.steps, a.warmup, world)
  if world > 1:
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
  total_env_steps = a.steps * n * T * world
  value = total_env_steps / dt
  assert bool(out[2][-1].all()) and not bool(out[2][:-1].any())     # done fires exactly at the horizon

  res = None
  if rank == 0:
    kmean = sum(kern_ms) / len(kern_ms)
    kmed = kern_ms[len(kern_ms) // 2]
    bytes_per_launch = n * (T * BYTES_PER_ENV_STEP_ROLLOUT + STATE_BYTES_PER_ENV_LAUNCH)
    achieved = bytes_per_launch / (kmean * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
      tj = json.load(open(tpath))
      key = f'rollout_n{n}_T{T}'
      traffic = tj.get(key, {}).get('hbm_bytes_per_launch')
    res = {
        'metric': 'env steps/sec (aggregate) at N parallel envs', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt / a.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'tabletop_manipulation {a.reward} reward, {n} batched envs per MI355X, '
                               f'reset + fused {T}-step rollout per bench step', 'envs_per_gpu': n,
                   'global_envs': n * world, 'horizon': T, 'env_steps_per_bench_step': n * T * world,
                   'parallelism': f'env-range shard x{world}, no per-step collective'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'kernel': 'rollout_ws_kernel',
                     'kernel_ms_mean': kmean, 'kernel_ms_median': kmed, 'algorithmic_bytes_per_launch': bytes_per_launch,
                     'bytes_per_env_step': BYTES_PER_ENV_STEP_ROLLOUT},
    }
    if not a.no_step_api:
      env2 = make_env(torch, n, T, a.reward, rank, device)
      ks = max(1, a.steps // 20)
      sdt, sgpu = time_step_api(torch, env2, acts, ks, a.warmup)
      res['step_api'] = {'value': ks * n * T / sdt, 'unit': 'env-steps/s', 'launches': ks * T,
                         'us_per_step_call_wall': sdt / (ks * T) * 1e6, 'us_per_step_call_gpu': sgpu / (ks * T) * 1e6,
                         'note': 'same workload, one eager step() launch per env step (host-launch bound at N=4096)',
                         'achieved_GBs': n * BYTES_PER_ENV_STEP_STEP / (sgpu / (ks * T)) / 1e9}
    if a.sweep:
      sw = []
      for ns in (64, 1024, 4096, 16384, 65536, 262144, 1048576):
        Ts = T if ns * T * 66 < 12e9 else max(8, int(12e9 // (ns * 66)))
        e = make_env(torch, ns, Ts, a.reward, 0, device)
        ac = synth_actions(torch, Ts, ns, 0, device)
        o = alloc_out(torch, Ts, ns, device)
        k = max(3, min(a.steps, int(2e9 // (ns * Ts * 66)) + 3))
        d, km, _ = time_rollouts(torch, dist, e, ac, o, k, 3, 1)
        kmn = sum(km) / len(km)
        sw.append({'envs': ns, 'T': Ts, 'env_steps_per_s': k * ns * Ts / d, 'kernel_ms': kmn,
                   'GBs': ns * (Ts * 66 + STATE_BYTES_PER_ENV_LAUNCH) / (kmn * 1e-3) / 1e9})
        del e, ac, o
        torch.cuda.empty_cache()
      res['sweep'] = sw
    if not a.no_cpu:
      res['cpu_baseline'] = cpu_baseline(n, T, a.reward, a.cpu_seconds)
    else:
      res['cpu_baseline'] = None
    print(json.dumps(res), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
