"""Throughput of the two Sawyer envs vs the number of envs (reset + fused rollout per step, random actions):
python tools/sweep_sawyer.py > profiles/r01_sawyer_n_sweep.txt   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg

print('workgroups: door 4 envs (one wave), four per CU; peg 16 envs (four waves), one per CU; 256 CUs')
for name, cls, T, per_cu in (('sawyer_door', SawyerDoor, 300, 16), ('sawyer_peg', SawyerPeg, 200, 16)):
  for n in (256, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768, 65536):
    iters = 3 if n <= 16384 else 1
    env = cls(num_envs=n)
    T_ = T                                  # full horizon at every N: the first steps after a reset are contact-free and cheaper
    acts = torch.rand(T_, n, 4, device='cuda') * 2 - 1
    out = env._new_out((T_,))
    env.reset(); env.rollout(acts, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      env.reset(); env.rollout(acts, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f'{name:12s} N={n:6d} T={T_:3d}: {ms:8.2f} ms, {n * T_ / ms / 1e3:7.2f} M env-steps/s, {n / (256 * per_cu):5.2f} rounds of resident envs')
    del env, acts, out
