"""minitaur_duo_kernel (two waves per SIMD by role) against the one-wave kernel: same outputs? how fast?   python tools/mt_duo_check.py [N] [T]   (GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.minitaur import Minitaur
lib = _abi.load()
nums = [int(x) for x in sys.argv[1:] if x.isdigit()]
n, T = (nums + [4096, 200])[:2] if len(nums) < 2 else nums[:2]
res = {}
for mode in (0, 1):
  lib.earl_debug_set_minitaur_duo(mode)
  env = Minitaur(num_envs=n, seed=1234, scalar_api=False)
  g = torch.Generator(device='cuda').manual_seed(99)
  acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1).float()
  env.reset(); r = env.rollout(acts)
  torch.cuda.synchronize()
  reps = 2
  t0 = time.perf_counter()
  for _ in range(reps):
    env.reset(); r = env.rollout(acts)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  res[mode] = {k: v.clone() for k, v in r.items()}
  res[mode]['qpos'], res[mode]['qvel'] = env.qpos.clone(), env.qvel.clone()
  print(f'duo={mode} minitaur N={n} T={T}: {dt * 1e3:8.2f} ms per reset + rollout, {n * T / dt / 1e6:7.2f} M env-steps/s, obs checksum {float(r["obs"].sum()):.12e}, failed env steps {int(env.fail_count.sum())}', flush=True)
  del env
lib.earl_debug_set_minitaur_duo(0)
for k in res[0]:
  a, b = res[0][k], res[1][k]
  same = torch.equal(a, b)
  d = (a.double() - b.double()).abs()
  first = None
  if not same and a.dim() >= 2 and a.shape[0] == T:
    bad = (d.reshape(T, -1) > 0).any(1).nonzero()
    first = int(bad[0]) if len(bad) else None
  print(f'  {k:8s} identical {same}   max |diff| {float(d.max()):.3e}' + (f'   first differing env step {first}' if first is not None else ''))
