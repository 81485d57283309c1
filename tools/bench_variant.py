"""Time the Sawyer door bench step (reset + 300-step rollout, N = 8192) through alternative builds of the library
(tools/ubench/libearl_<tag>.so, built with -DEARL_DOOR_COOP / -DEARL_DOOR_WPB): python tools/bench_variant.py <tag> [peg]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
if sys.argv[1] != 'ship':
  _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_{sys.argv[1]}.so')
peg = len(sys.argv) > 2 and sys.argv[2] == 'peg'
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
from earl_benchmark_amd.wrappers import PersistentStateWrapper
n, T = 8192, 200 if peg else 300
env = PersistentStateWrapper((SawyerPeg if peg else SawyerDoor)(num_envs=n, seed=1234), T)
g = torch.Generator(device='cuda').manual_seed(99)
acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
out = env.unwrapped._new_out((T,))
for _ in range(2):
  env.reset(); env.rollout(acts, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
  env.reset(); env.rollout(acts, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
print(f'{sys.argv[1]:8s} {"peg" if peg else "door"}: {dt * 1e3:8.2f} ms per rollout, {n * T / dt / 1e6:7.2f} M env-steps/s, obs checksum {float(out["obs"].sum()):.9e}')
