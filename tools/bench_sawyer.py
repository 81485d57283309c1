"""Sawyer door env rollout throughput. Usage: python tools/bench_sawyer.py [N] [T] [iters] [lanes_per_env]"""
import sys
import time

import torch

from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
  T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
  iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
  lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 16
  _abi.check(_abi.load().earl_debug_set_physics_lanes(lanes), 'set lanes')
  env = SawyerDoor(num_envs=n)
  acts = torch.rand(T, n, 4, device='cuda') * 2 - 1
  out = env._new_out((T,))
  for _ in range(2):
    env.reset(); env.rollout(acts, out=out)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    env.reset(); env.rollout(acts, out=out)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / iters
  print(f'N={n} T={T} lanes/env={lanes}: {ms:.2f} ms per reset+rollout, {n * T / ms / 1e3:.2f} M env-steps/s, '
        f'{n * T * 5 / ms / 1e3:.1f} M timesteps/s')


if __name__ == '__main__':
  main()
