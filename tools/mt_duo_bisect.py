"""where do the one-wave and the two-wave minitaur kernels part? one env step from identical states, per env   (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.minitaur import Minitaur
lib = _abi.load()
n = 4096
lib.earl_debug_set_minitaur_duo(0)
a = Minitaur(num_envs=n, seed=1234, scalar_api=False); b = Minitaur(num_envs=n, seed=1234, scalar_api=False)
a.reset(); b.reset()
g = torch.Generator(device='cuda').manual_seed(99)
tot = 0
for t in range(12):
  act = (torch.rand(1, n, 8, generator=g, device='cuda') * 2 - 1).float()
  for k in ('qpos', 'qvel', 'overheat', 'motor_enabled', 'observed_torque', 'steps_since_reset'):
    getattr(b, k).copy_(getattr(a, k))
  lib.earl_debug_set_minitaur_duo(0); ra = a.rollout(act)
  lib.earl_debug_set_minitaur_duo(1); rb = b.rollout(act)
  torch.cuda.synchronize()
  dq = (a.qvel - b.qvel).abs()
  bad = (dq.max(1).values > 0).nonzero().flatten()
  tot += len(bad)
  msg = f'step {t}: {len(bad)} of {n} envs differ'
  if len(bad):
    e = int(bad[0])
    msg += f'; env {e}: base z {float(a.qpos[e, 2]):.4f}, |qvel diff| per dof ' + ' '.join(f'{float(x):.1e}' for x in dq[e]) + f'; differing envs mod 16: {sorted(set((bad % 16).tolist()))}'
  print(msg, flush=True)
print('total', tot)
