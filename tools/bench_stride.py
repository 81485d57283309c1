"""Time the three articulated-body workloads through an alternative build of the library (tools/ubench/libearl_<tag>.so; 'ship' = the product
build): door N = 8192 (eight-wave build) and N = 4096 (single-wave build) x 300 steps, peg N = 8192 x 200, kitchen N = 2048 x 20 env steps.
Used for the LDS block-stride sweep (csrc/physics.hip Shared<NV>::TARGET):  python tools/bench_stride.py <tag>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
tag = sys.argv[1]
if tag != 'ship':
  _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_{tag}.so')
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
from earl_benchmark_amd.envs.kitchen import Kitchen
from earl_benchmark_amd.wrappers import PersistentStateWrapper


def sawyer(cls, n, T):
  env = PersistentStateWrapper(cls(num_envs=n, seed=1234), T)
  g = torch.Generator(device='cuda').manual_seed(99)
  acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
  out = env.unwrapped._new_out((T,))
  env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / 3 * 1e3, float(out['obs'].sum())


def kitchen(n, T):
  env = Kitchen(num_envs=n, seed=5)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(7)
  acts = (torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1).float()
  for t in range(3):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(T):
    o = env.step(acts[t])[0]
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / T * 1e3, float(o.sum())


d8, c1 = sawyer(SawyerDoor, 8192, 300)
d4, c2 = sawyer(SawyerDoor, 4096, 300)
p8, c3 = sawyer(SawyerPeg, 8192, 200)
k, c4 = kitchen(2048, 20)
print(f'{tag:12s} door8192 {d8:7.2f} ms  door4096 {d4:7.2f} ms  peg8192 {p8:7.2f} ms  kitchen {k:6.3f} ms/env-step   checksums {c1:.6e} {c2:.6e} {c3:.6e} {c4:.6e}', flush=True)
