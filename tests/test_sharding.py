"""N > 1 path on CPU: env-range partition + the one collective, world size 2 over gloo (two real processes)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO
from earl_benchmark_amd import sharding


def test_shard_ranges_partition_the_envs():
  for n in (1, 7, 64, 4096, 4097, 100003):
    for w in (1, 2, 3, 4, 8):
      r = [sharding.shard_range(n, k, w) for k in range(w)]
      assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
      sizes = [b - a for a, b in r]
      assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
  with pytest.raises(ValueError):
    sharding.shard_range(8, 2, 2)
  assert sharding.shard_kwargs(4096, rank=3, world_size=8) == {'num_envs': 512, 'env_offset': 1536}


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _worker(rank, world, port, n_global, T, ragged, out_dir):
  sys.path.insert(0, REPO)
  import torch.distributed as dist
  from earl_benchmark_amd import sharding as sh
  from oracle import tabletop_oracle as orc       # the checker plays the env here: the product path is GPU-only
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    kw = sh.shard_kwargs(n_global + (1 if ragged else 0))
    lo, n = kw['env_offset'], kw['num_envs']
    rng = np.random.default_rng(0)
    acts_global = rng.uniform(-1, 1, size=(T, n_global + (1 if ragged else 0), 3)).astype(np.float32)
    o = orc.OracleTabletop(n, horizon=T, seed=5, env_offset=lo)
    o.reset()
    obs, rew, done, succ = o.rollout(np.ascontiguousarray(acts_global[:, lo:lo + n]))
    summary = sh.rollout_summary(torch.from_numpy(rew), torch.from_numpy(succ.astype(bool)))
    table = sh.gather_summary(summary)
    # the same gather when the caller supplies the (deterministic) shard sizes: no size exchange
    known = [b - a for a, b in (sh.shard_range(n_global + (1 if ragged else 0), r, world) for r in range(world))]
    assert torch.equal(sh.gather_summary(summary, sizes=known), table)
    # max-over-ranks timing plumbing of bench.py: all_reduce(MAX)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    np.save(os.path.join(out_dir, f'table_{rank}.npy'), table.numpy())
    np.save(os.path.join(out_dir, f'goal_{rank}.npy'), o.goal_idx)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('ragged', [False, True])
def test_two_rank_gather_equals_single_batch(tmp_path, ragged):
  from oracle import tabletop_oracle as orc
  n_global, T, world = 96, 25, 2
  port = _free_port()
  mp.spawn(_worker, args=(world, port, n_global, T, ragged, str(tmp_path)), nprocs=world, join=True)
  n = n_global + (1 if ragged else 0)
  rng = np.random.default_rng(0)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  o = orc.OracleTabletop(n, horizon=T, seed=5)
  o.reset()
  obs, rew, done, succ = o.rollout(acts)
  want = np.stack([rew.sum(0), succ[-1].astype(np.float32)], 1)
  for r in range(world):
    np.testing.assert_array_equal(np.load(tmp_path / f'table_{r}.npy'), want)   # every rank holds the full table
  goals = np.concatenate([np.load(tmp_path / f'goal_{r}.npy') for r in range(world)])
  np.testing.assert_array_equal(goals, o.goal_idx)                               # RNG keyed by the global env id
