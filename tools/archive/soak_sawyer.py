import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
for cls, kw, n in ((SawyerDoor, {}, 4096), (SawyerDoor, {}, 8192), (SawyerPeg, {}, 4096), (SawyerPeg, dict(reset_at_goal=True), 4096), (SawyerPeg, dict(wide_init=True), 4096)):
  T = 500                                               # door at 8192 envs = the eight-waves-per-CU build
  env = cls(num_envs=n, seed=3, **kw)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(1)
  worst = 0.0
  for rep in range(6):                                  # 3000 env steps without a reset
    acts = torch.rand(T, n, 4, device='cuda', generator=g) * 2 - 1
    if rep % 2: acts[..., 2] = -acts[..., 2].abs()       # press down half of the time
    out = env.rollout(acts)
    obs = out['obs']
    assert torch.isfinite(obs).all() and torch.isfinite(env.qpos).all() and torch.isfinite(env.qvel).all(), (cls.__name__, kw, rep)
    worst = max(worst, float(env.qvel.abs().max()))
    lo = obs[..., :3].amin((0, 1)).cpu().numpy(); hi = obs[..., :3].amax((0, 1)).cpu().numpy()
    olo = obs[..., 4:7].amin((0, 1)).cpu().numpy(); ohi = obs[..., 4:7].amax((0, 1)).cpu().numpy()
  print(cls.__name__, kw, 'ok: 3000 steps x', n, 'envs; guard trips', int(env.fail_count.sum()), '; max |qvel| %.1f' % worst, 'hand range', lo.round(3), hi.round(3), 'object range', olo.round(3), ohi.round(3), flush=True)
