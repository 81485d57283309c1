"""Peg bench step (reset + 200-step rollout, N = 8192) under the schedules of earl_debug_set_peg_schedule: 0 = one env group per wave (static, two rounds),
k >= 2 = time-sliced work queue with k env steps per item.   python tools/bench_peg_schedule.py [k ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
from earl_benchmark_amd.wrappers import PersistentStateWrapper
n, T = 8192, 200
lib = _abi.load()
g = torch.Generator(device='cuda').manual_seed(99)
acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
ref = None
for k in [int(x) for x in sys.argv[1:]] or [0, 5, 10, 20, 40, 100]:
  lib.earl_debug_set_peg_schedule(k)
  env = PersistentStateWrapper(SawyerPeg(num_envs=n, seed=1234), T)      # (a fresh env per schedule: the same Philox counters, hence the same episodes)
  out = env.unwrapped._new_out((T,))
  for _ in range(2):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(4):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 4
  chk = out['obs'].clone()
  same = True if ref is None else bool(torch.equal(chk, ref))
  ref = chk if ref is None else ref
  print(f'schedule {k:3d}: {dt * 1e3:7.2f} ms per reset + rollout = {n * T / dt / 1e6:6.2f} M env-steps/s; outputs identical to the first schedule: {same}')
lib.earl_debug_set_peg_schedule(1)
