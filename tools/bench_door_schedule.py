"""Door bench step (reset + 300-step rollout, N = 8192) under earl_debug_set_door_variant: 0 / 2 = eight waves per CU (shipped for this size), 1 = single-wave
workgroups, two rounds, 3 = single-wave workgroups under the time-sliced work queue of the peg.   python tools/bench_door_schedule.py [variant ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.wrappers import PersistentStateWrapper
n, T = 8192, 300
lib = _abi.load()
g = torch.Generator(device='cuda').manual_seed(99)
acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
ref = None
for k in [int(x) for x in sys.argv[1:]] or [2, 1, 3]:
  lib.earl_debug_set_door_variant(k)
  env = PersistentStateWrapper(SawyerDoor(num_envs=n, seed=1234), T)
  env.unwrapped.door_queue = k == 3
  out = env.unwrapped._new_out((T,))
  for _ in range(2):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(4):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 4
  env2 = PersistentStateWrapper(SawyerDoor(num_envs=n, seed=1234), T)      # (fresh env: the same Philox counters for the identity check)
  env2.unwrapped.door_queue = k == 3
  env2.reset(); o2 = env2.rollout(acts)['obs'].clone()
  same = True if ref is None else bool(torch.equal(o2, ref))
  ref = o2 if ref is None else ref
  print(f'door variant {k}: {dt * 1e3:7.2f} ms per reset + rollout = {n * T / dt / 1e6:6.2f} M env-steps/s; outputs identical to the first variant: {same}')
lib.earl_debug_set_door_variant(0)
