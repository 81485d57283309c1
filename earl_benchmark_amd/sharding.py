"""Multi-GPU layout of the batched envs: contiguous env-id ranges per rank, no per-step communication, and one
collective (RCCL all-gather over xGMI; gloo on CPU in the tests) for the result of an evaluation rollout.

The reference has no parallelism at all; envs are independent, so sharding is a pure partition.
RNG streams are keyed by the GLOBAL env id (`env_offset + i`), which makes W shards bit-identical to one batch.
"""
import torch
import torch.distributed as dist


def shard_range(num_envs_global, rank, world_size):
  """[start, stop) of the envs owned by `rank`: contiguous, sizes differ by at most one, lower ranks get the extras."""
  if not (0 <= rank < world_size):
    raise ValueError(f'rank {rank} not in [0, {world_size})')
  base, extra = divmod(int(num_envs_global), int(world_size))
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def shard_kwargs(num_envs_global, rank=None, world_size=None):
  """kwargs (`num_envs`, `env_offset`) for EARLEnvs / TabletopManipulation on this rank."""
  if rank is None:
    rank = dist.get_rank() if dist.is_initialized() else 0
  if world_size is None:
    world_size = dist.get_world_size() if dist.is_initialized() else 1
  lo, hi = shard_range(num_envs_global, rank, world_size)
  return {'num_envs': hi - lo, 'env_offset': lo}


def rollout_summary(reward, success):
  """[n, 2] float32 per-env result of an evaluation rollout: undiscounted return and success at the last step.
  reward [T, n] float32, success [T, n] bool."""
  return torch.stack([reward.sum(0), success[-1].to(reward.dtype)], 1).contiguous()


def gather_summary(summary, group=None, sizes=None):
  """The single collective of an evaluation job: every rank receives the [N_global, 2] table (rank order = env order).
  One all_gather_into_tensor of the (padded, if the shards are ragged) per-rank tables.  `sizes` (rows per rank), when
  the caller knows them -- shard_range is deterministic -- saves the size exchange and its host synchronisation."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return summary
  world = dist.get_world_size(group)
  if sizes is None:
    sizes = [torch.zeros(1, dtype=torch.int64, device=summary.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([summary.shape[0]], dtype=torch.int64, device=summary.device), group=group)
    sizes = [int(s.item()) for s in sizes]
  assert len(sizes) == world and sizes[dist.get_rank(group)] == summary.shape[0]
  if len(set(sizes)) == 1:
    out = torch.empty(world * sizes[0], *summary.shape[1:], dtype=summary.dtype, device=summary.device)
    dist.all_gather_into_tensor(out, summary.contiguous(), group=group)
    return out
  # ragged shards (sizes differ by one): pad to the largest, gather once, drop the padding
  m = max(sizes)
  padded = torch.zeros(m, *summary.shape[1:], dtype=summary.dtype, device=summary.device)
  padded[:summary.shape[0]] = summary
  out = torch.empty(world * m, *summary.shape[1:], dtype=summary.dtype, device=summary.device)
  dist.all_gather_into_tensor(out, padded, group=group)
  return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], 0)


def pack_rollout(obs, reward, done, success):
  """[T, n, D + 2] float32 trajectory buffer of one rank: the observation, the reward, and `done | success << 1`
  as one word (small integers are exact in float32)."""
  flags = (done.to(torch.int32) | (success.to(torch.int32) << 1)).to(torch.float32)
  return torch.cat([obs.to(torch.float32), reward.to(torch.float32)[..., None], flags[..., None]], -1).contiguous()


def unpack_rollout(buf):
  """-> obs [T, N, D], reward [T, N], done [T, N] bool, success [T, N] bool"""
  flags = buf[..., -1].to(torch.int32)
  return buf[..., :-2], buf[..., -2], (flags & 1).bool(), ((flags >> 1) & 1).bool()


def gather_rollout(buf, group=None, sizes=None):
  """The trajectory-collecting variant of the job's one collective: every rank receives [T, N_global, D + 2], env order = rank order
  (one all_gather_into_tensor of the per-rank [T, n, D + 2] buffers, padded along the env axis if the shards are ragged).
  Message per rank: T * n * (D + 2) * 4 bytes (tabletop T = 200, n = 4096: 45.9 MB) -- over xGMI a direct all-gather is seven
  concurrent point-to-point sends of one shard each, per-link bound at shard / 153 GB/s."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return buf
  world = dist.get_world_size(group)
  if sizes is None:
    sz = [torch.zeros(1, dtype=torch.int64, device=buf.device) for _ in range(world)]
    dist.all_gather(sz, torch.tensor([buf.shape[1]], dtype=torch.int64, device=buf.device), group=group)
    sizes = [int(x.item()) for x in sz]
  assert len(sizes) == world and sizes[dist.get_rank(group)] == buf.shape[1]
  T, m = buf.shape[0], max(sizes)
  if buf.shape[1] != m:
    padded = torch.zeros(T, m, buf.shape[2], dtype=buf.dtype, device=buf.device)
    padded[:, :buf.shape[1]] = buf
    buf = padded
  out = torch.empty(world * T, m, buf.shape[2], dtype=buf.dtype, device=buf.device)     # rank-major along dim 0 (what gloo / RCCL both accept)
  dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
  out = out.view(world, T, m, buf.shape[2])
  return torch.cat([out[r, :, :sizes[r]] for r in range(world)], 1)
