"""bench.py's own N > 1 sequence on CPU: two real processes over gloo run `bench.time_rollouts` -- warm-up, barrier, the timed loop,
the job's ONE collective (`gather_summary`, optionally `gather_rollout`), barrier, all_reduce(MAX) of the wall time -- exactly the code
the 8-GPU scaling run executes over RCCL.  There is no GPU here, so the env's launch is replaced by the HIP library's own ARGUMENT
VALIDATOR: the real C-ABI entry point `earl_tabletop_reset_rollout` called with `cfg.n = 0` (every pointer and field is checked, then it
returns EARL_OK before any launch; a broken argument makes it fail the test).  The outputs the missing kernel would have written are
filled with a deterministic function of the GLOBAL env id, so the gathered tables can be checked entry by entry.  The oracle is not
involved: this test is about the plumbing, not the arithmetic."""
import ctypes as C
import json
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def synthetic(T, lo, n, shift=0):
  """what the stub 'kernel' writes for global envs [lo, lo + n): obs, reward, done, success (`shift`: a different table per episode)"""
  gid = torch.arange(lo, lo + n, dtype=torch.float32) + 1000.0 * shift
  t = torch.arange(T, dtype=torch.float32)[:, None]
  obs = (gid[None, :, None] * 0.5 + t[..., None] + torch.arange(12, dtype=torch.float32) * 0.25).contiguous()
  reward = ((gid[None, :] + t) % 3 == 0).to(torch.float32)
  done = (t == T - 1).expand(T, n).contiguous()
  success = ((gid[None, :].to(torch.int64) + t.to(torch.int64)) % 5 == 0)
  return obs, reward, done, success


class ValidatorEnv:
  """host tensors with the product's state layout; rollout() = the library's argument validation + synthetic outputs"""

  def __init__(self, n, T, lo, delay):
    from earl_benchmark_amd import _abi
    self.lib, self._abi = _abi.load(), _abi
    self.n, self.T, self.lo, self.delay, self.launches = n, T, lo, delay, 0
    self.qpos = torch.zeros(n, 4, dtype=torch.float64)
    self.attached = torch.zeros(n, dtype=torch.int8)
    self.goal_idx = torch.zeros(n, dtype=torch.int32)
    self.goal_table = torch.zeros(4, 6, dtype=torch.float64)
    self.i32 = [torch.zeros(n, dtype=torch.int32) for _ in range(3)]
    self.lret = torch.zeros(n, dtype=torch.float64)
    self.cfg = _abi.TabletopCfg(n=0, env_offset=lo, reward_type=0, wide_init=0, reset_at_goal=0, horizon=T, goal_change_frequency=0,
                                auto_reset=0, n_goals=4, n_sample_goals=4, seed=0, counter=0)      # n = 0: validate, do not launch
    self.st = _abi.TabletopState(self.qpos.data_ptr(), self.attached.data_ptr(), self.goal_idx.data_ptr(), self.goal_table.data_ptr(),
                                 self.i32[0].data_ptr(), self.i32[1].data_ptr(), self.i32[2].data_ptr(), self.lret.data_ptr())

  def _check_set(self, acts):                              # launch j of the job reads action tensor j % R
    sets = getattr(self, 'expect_sets', None)
    assert sets is None or acts is sets[self.launches % len(sets)]

  def rollout(self, acts, out, reset_first):
    self._check_set(acts)
    assert reset_first and acts.shape == (self.T, self.n, 3) and acts.dtype == torch.float32 and acts.is_contiguous()
    o = self._abi.TabletopOut(*(t.data_ptr() for t in out))
    rc = self.lib.earl_tabletop_reset_rollout(C.byref(self.cfg), C.byref(self.st), self.T, acts.data_ptr(), C.byref(o), None)
    self._abi.check(rc, 'earl_tabletop_reset_rollout (argument validation)')
    assert self.lib.earl_tabletop_reset_rollout(C.byref(self.cfg), C.byref(self.st), self.T, None, C.byref(o), None) != 0   # it does validate
    for dst, src in zip(out, synthetic(self.T, self.lo, self.n)):
      dst.copy_(src)
    self.launches += 1
    time.sleep(self.delay)


  def rollout_episodes(self, acts, out, episodes=None):
    self._check_set(acts)
    E = acts.shape[0]
    assert acts.shape == (E, self.T, self.n, 3) and acts.is_contiguous() and all(t.shape[0] == E for t in out) and episodes in (None, E)
    assert not torch.equal(acts[0], acts[-1])                      # every episode of the launch has its own actions
    o = self._abi.TabletopOut(*(t.data_ptr() for t in out))
    stride = self.T * self.n * 3
    rc = self.lib.earl_tabletop_eval_episodes(C.byref(self.cfg), C.byref(self.st), E, self.T, acts.data_ptr(), stride, C.byref(o), None)
    self._abi.check(rc, 'earl_tabletop_eval_episodes (argument validation)')
    assert self.lib.earl_tabletop_eval_episodes(C.byref(self.cfg), C.byref(self.st), -1, self.T, acts.data_ptr(), stride, C.byref(o), None) != 0
    assert self.lib.earl_tabletop_eval_episodes(C.byref(self.cfg), C.byref(self.st), E, self.T, acts.data_ptr(), -1, C.byref(o), None) != 0
    for e in range(E):                                             # episode e's rows: the synthetic table shifted by e (the LAST one is gathered)
      for dst, src in zip(out, synthetic(self.T, self.lo, self.n, shift=E - 1 - e)):
        dst[e].copy_(src)
    self.launches += 1
    self.episodes = getattr(self, 'episodes', 0) + E
    time.sleep(self.delay)


def _worker(rank, world, port, n, T, steps, warmup, out_dir, E=1):
  sys.path.insert(0, REPO)
  import torch.distributed as dist
  import bench
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    env = ValidatorEnv(n, T, rank * n, delay=0.02 * (rank + 1))        # rank 1 is the slow one: the job time is ITS time
    # [E, T, n, 3] distinct per episode (E > 1), two such tensors read round-robin by the launches, as bench.main() feeds them
    acts = [bench.synth_actions(torch, T, n, rank + 1000 * r, 'cpu', E) for r in range(2)]
    env.expect_sets = acts
    out = bench.alloc_out(torch, T, n, 'cpu', E)
    dt, kern_ms, table, traj, launches = bench.time_rollouts(torch, dist, env, acts, out, steps, warmup, world, device='cpu', gather_rollout=True)
    assert launches == steps and env.launches == steps + warmup       # a bench step is ONE launch, whatever E
    assert E == 1 or env.episodes == (steps + warmup) * E
    np.save(os.path.join(out_dir, f'table_{rank}.npy'), table.numpy())
    np.save(os.path.join(out_dir, f'traj_{rank}.npy'), traj.numpy())
    json.dump({'dt': dt, 'kern_ms': kern_ms}, open(os.path.join(out_dir, f'time_{rank}.json'), 'w'))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('E', [1, 3])
def test_two_rank_bench_sequence(tmp_path, E):
  from earl_benchmark_amd import sharding
  world, n, T, steps, warmup = 2, 48, 7, 3, 2
  mp.spawn(_worker, args=(world, _free_port(), n, T, steps, warmup, str(tmp_path), E), nprocs=world, join=True)
  obs, reward, done, success = synthetic(T, 0, world * n)
  want = sharding.rollout_summary(reward, success).numpy()
  times = [json.load(open(tmp_path / f'time_{r}.json')) for r in range(world)]
  for r in range(world):
    np.testing.assert_array_equal(np.load(tmp_path / f'table_{r}.npy'), want)            # every rank holds the whole [N, 2] table, env order
    o, rw, d, s = sharding.unpack_rollout(torch.from_numpy(np.load(tmp_path / f'traj_{r}.npy')))
    assert o.shape == (T, world * n, 12)
    assert torch.equal(o, obs) and torch.equal(rw, reward) and torch.equal(d, done) and torch.equal(s, success)
  assert times[0]['dt'] == times[1]['dt'] >= steps * 0.04                                  # MAX over ranks: the slow rank's time, on both
  assert times[0]['kern_ms'][0] < times[1]['kern_ms'][0]                                   # ... while the per-rank launch clock stays local


def test_single_process_sequence_and_ragged_gather_rollout():
  """world size 1 runs the same function without a process group; gather_rollout of ragged shards pads and trims (2 threads as ranks
  would need a group: the ragged arithmetic is checked through the single-process identity and the packing round trip)"""
  import bench
  from earl_benchmark_amd import sharding
  n, T = 40, 5
  env = ValidatorEnv(n, T, 0, delay=0.0)
  out = bench.alloc_out(torch, T, n, 'cpu')
  dt, kern_ms, table, traj, _ = bench.time_rollouts(torch, None, env, bench.synth_actions(torch, T, n, 0, 'cpu'), out, 2, 1, 1, device='cpu')
  assert table is None and traj is None and dt > 0 and env.launches == 3
  buf = sharding.pack_rollout(*out)
  assert buf.shape == (T, n, 14) and sharding.gather_rollout(buf) is buf
  o, rw, d, s = sharding.unpack_rollout(buf)
  for got, ref in zip((o, rw, d, s), synthetic(T, 0, n)):
    assert torch.equal(got, ref)


@pytest.mark.parametrize('ragged_sizes', [(5, 4)])
def test_ragged_gather_rollout_two_ranks(tmp_path, ragged_sizes):
  mp.spawn(_ragged_worker, args=(2, _free_port(), ragged_sizes, str(tmp_path)), nprocs=2, join=True)
  T, lo = 3, 0
  full = []
  for r, m in enumerate(ragged_sizes):
    full.append(synthetic(T, lo, m))
    lo += m
  want = [torch.cat([f[k] for f in full], 1) for k in range(4)]
  for r in range(2):
    got = torch.from_numpy(np.load(tmp_path / f'ragged_{r}.npy'))
    from earl_benchmark_amd import sharding
    for a, b in zip(sharding.unpack_rollout(got), want):
      assert torch.equal(a, b)


def _ragged_worker(rank, world, port, sizes, out_dir):
  sys.path.insert(0, REPO)
  import torch.distributed as dist
  from earl_benchmark_amd import sharding
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    lo = sum(sizes[:rank])
    buf = sharding.pack_rollout(*synthetic(3, lo, sizes[rank]))
    a = sharding.gather_rollout(buf)                       # sizes exchanged
    b = sharding.gather_rollout(buf, sizes=list(sizes))    # sizes known
    assert torch.equal(a, b)
    np.save(os.path.join(out_dir, f'ragged_{rank}.npy'), a.numpy())
  finally:
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------------------------------
# bench.run_kitchen / bench.run_minitaur with world = 2: the STRONG-scaling lines -- a fixed global batch range-sharded over the
# ranks, reset + ONE fused launch per bench step, the job's one collective (gather_summary of the last episode), barrier, MAX over ranks.  The env is a CPU
# stand-in with the product env's surface (num_envs / env_offset constructor, _cfg.horizon set by the wrapper, reset(), rollout(acts, out=...), fail_count)
# whose "launch" writes a deterministic function of the GLOBAL env id: the plumbing is under test, not the arithmetic.
class _Cfg:
  horizon = 0


class StandInEnv:
  OBS_DIM = 5
  scalar_api = False

  def __init__(self, num_envs=1, seed=0, env_offset=0, scalar_api=False, obs_dim=5, act_dim=9):
    self.num_envs, self.env_offset, self.OBS_DIM, self.act_dim = num_envs, env_offset, obs_dim, act_dim
    self._cfg = _Cfg()
    self.fail_count = torch.zeros(num_envs, dtype=torch.int32)
    self.resets = self.launches = 0
    self.delay = 0.0

  @property
  def unwrapped(self):
    return self

  def reset(self, mask=None):
    self.resets += 1
    return torch.zeros(self.num_envs, self.OBS_DIM, dtype=torch.float64)

  def _new_out(self, lead):
    return {}

  def rollout(self, actions, out=None):
    T, n = actions.shape[0], self.num_envs
    assert actions.shape == (T, n, self.act_dim) and actions.dtype == torch.float32 and T == self._cfg.horizon
    res = out if out is not None else {}
    gid = torch.arange(self.env_offset, self.env_offset + n, dtype=torch.float64)
    t = torch.arange(T, dtype=torch.float64)[:, None]
    res['obs'] = (gid[None, :, None] + t[..., None] * 0.5 + torch.zeros(self.OBS_DIM, dtype=torch.float64)).contiguous()
    res['reward'] = -(gid[None, :] * 0.25 + t)
    res['done'] = (t == T - 1).expand(T, n).contiguous()
    res['success'] = ((gid[None, :].to(torch.int64) + t.to(torch.int64)) % 7 == 0)
    res['status'] = torch.zeros(T, n, dtype=torch.uint8)
    self.launches += 1
    time.sleep(self.delay)
    return res


def _strong_worker(rank, world, port, which, n_global, T, steps, warmup, out_dir):
  sys.path.insert(0, REPO)
  import argparse
  import torch.distributed as dist
  import bench
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    made = []

    def factory(**kw):
      e = StandInEnv(act_dim=9 if which == 'kitchen' else 8, **kw)
      e.delay = 0.02 * (rank + 1)                                     # rank 1 is the slow one
      made.append(e)
      return e
    a = argparse.Namespace(no_step_api=True)
    fn = bench.run_kitchen if which == 'kitchen' else bench.run_minitaur
    res = fn(a, torch, dist, world, rank, 'cpu', steps, warmup, n_global=n_global, T=T, cpu_seconds=None, env_factory=factory)
    env = made[0]
    lo, hi = __import__('earl_benchmark_amd.sharding', fromlist=['x']).shard_range(n_global, rank, world)
    assert env.num_envs == hi - lo and env.env_offset == lo and env._cfg.horizon == T          # its own contiguous range, the wrapper's horizon
    assert env.launches == steps + warmup and env.resets == steps + warmup                      # a bench step = reset + ONE launch
    assert (res is None) == (rank != 0)
    if rank == 0:
      json.dump({k: res[k] for k in ('value', 'ms_per_step', 'scaling', 'gathered_rows', 'config')}, open(os.path.join(out_dir, f'{which}.json'), 'w'))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('which, n_global', [('kitchen', 21), ('minitaur', 32)])      # (21: ragged shards, 11 + 10)
def test_two_rank_strong_scaling_lines(tmp_path, which, n_global):
  import bench
  world, T, steps, warmup = 2, 6, 2, 1
  mp.spawn(_strong_worker, args=(world, _free_port(), which, n_global, T, steps, warmup, str(tmp_path)), nprocs=world, join=True)
  res = json.load(open(tmp_path / f'{which}.json'))
  assert res['scaling'] == 'strong' and res['gathered_rows'] == n_global                       # every rank held the whole [N_global, 2] summary
  assert res['ms_per_step'] >= 40.0 - 1e-6                                                      # MAX over ranks: the slow rank's 2 x 20 ms per step
  assert abs(res['value'] - n_global * T / (res['ms_per_step'] * 1e-3)) < 1e-6 * res['value']  # whole-job env steps over the max-over-ranks time
  cfg = res['config']
  assert cfg['envs_global'] == n_global and cfg['envs_per_gpu'] == -(-n_global // world) and 'strong scaling' in cfg['parallelism']
  # what the kernels' layout predicts for the real shapes (one round of 2048 resident envs per GPU): stated in the line
  # the line says what the kernels' layout gives, from MEASURED shard launches (profiles/r05_kitchen_small_batch.txt), never 'x8'
  k8, m8 = bench.predicted_scaling('kitchen', 2048, 8), bench.predicted_scaling('minitaur', 4096, 8)
  assert 1.0 <= k8['strong']['predicted_speedup_vs_1_gpu'] < 1.9 and 1.5 <= m8['strong']['predicted_speedup_vs_1_gpu'] < 2.0
  for p8 in (k8, m8):                                            # static figures, labelled so, with the file they come from -- which exists and holds this workload
    assert p8['strong']['basis'].startswith('static') and bench.SHARD_PROFILE in p8['strong']['basis'] and os.path.exists(os.path.join(REPO, bench.SHARD_PROFILE))
  prof = open(os.path.join(REPO, bench.SHARD_PROFILE)).read()
  for wl, shard in (('kitchen', 256), ('minitaur', 512)):
    row = [ln for ln in prof.splitlines() if ln.startswith(wl + ': shard of 8 GPU(s)')][0]
    assert f'{shard} envs' in row and f'{bench.MEASURED_SHARD_TIME[wl][8]:.2f} x the' in row
  # ... and the WEAK-scaling entry next to it: the config's batch PER GPU, predicted `world` x -- the regime the design scales in
  assert k8['weak'] == {**k8['weak'], 'envs_per_gpu': 2048, 'envs_global': 16384, 'predicted_speedup_vs_1_gpu': 8.0} and m8['weak']['envs_global'] == 32768
  assert bench.predicted_scaling('kitchen', 8192, 2)['strong']['predicted_speedup_vs_1_gpu'] == 2.0       # other batch sizes: by launch rounds
  assert cfg['predicted_scaling']['strong']['envs_per_gpu'] == -(-n_global // world) and cfg['predicted_scaling']['weak']['envs_per_gpu'] == n_global


def _ragged_trajectory_worker(rank, world, port, n_global, T, D, out_dir):
  sys.path.insert(0, REPO)
  import torch.distributed as dist
  from earl_benchmark_amd import sharding
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    lo, hi = sharding.shard_range(n_global, rank, world)
    g = torch.Generator().manual_seed(5)
    obs = torch.randn(T, n_global, D, generator=g)[:, lo:hi].contiguous()              # this rank's columns of ONE global trajectory buffer
    g2 = torch.Generator().manual_seed(6)
    rew = torch.randn(T, n_global, generator=g2)[:, lo:hi].contiguous()
    done = (torch.arange(T)[:, None] == T - 1).expand(T, hi - lo).contiguous()
    succ = ((torch.arange(lo, hi)[None, :] + torch.arange(T)[:, None]) % 3 == 0)
    for sizes in (None, [sharding.shard_range(n_global, r, world)[1] - sharding.shard_range(n_global, r, world)[0] for r in range(world)]):
      full = sharding.gather_rollout(sharding.pack_rollout(obs, rew, done, succ), sizes=sizes)     # with and without the size exchange
      torch.save(full, os.path.join(out_dir, f'r{rank}_{0 if sizes is None else 1}.pt'))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world, n_global', [(2, 21), (3, 8)])      # ragged: 11 + 10, and 3 + 3 + 2
def test_gather_rollout_of_ragged_shards_reproduces_the_one_batch_trajectory_buffer(tmp_path, world, n_global):
  """the job's trajectory-collecting collective on shards whose sizes differ -- every rank ends up holding [T, N_global, D + 2] equal, entry by
  entry, to the buffer ONE batch of all the envs would have packed (env order = rank order; padding dropped)."""
  from earl_benchmark_amd import sharding
  T, D = 5, 12
  mp.spawn(_ragged_trajectory_worker, args=(world, _free_port(), n_global, T, D, str(tmp_path)), nprocs=world, join=True)
  obs = torch.randn(T, n_global, D, generator=torch.Generator().manual_seed(5))
  rew = torch.randn(T, n_global, generator=torch.Generator().manual_seed(6))
  done = (torch.arange(T)[:, None] == T - 1).expand(T, n_global).contiguous()
  succ = ((torch.arange(n_global)[None, :] + torch.arange(T)[:, None]) % 3 == 0)
  one_batch = sharding.pack_rollout(obs, rew, done, succ)
  for rank in range(world):
    for k in (0, 1):
      got = torch.load(tmp_path / f'r{rank}_{k}.pt')
      assert got.shape == (T, n_global, D + 2) and torch.equal(got, one_batch)
  o, r, d, sc = sharding.unpack_rollout(one_batch)
  assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d, done) and torch.equal(sc, succ)
