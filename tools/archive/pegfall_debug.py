import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
n, T = 512, 250
env = SawyerPeg(num_envs=n, seed=13); env.reset()
g = torch.Generator(device='cuda').manual_seed(17)
for c in range(8):
  acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
  q0 = env.qpos.clone(); v0 = env.qvel.clone(); m0 = env.mocap_pos.clone()
  out = env.rollout(acts)
  bad = (env.qpos[:, 11] < -0.02).nonzero().flatten()
  if len(bad):
    e = int(bad[0]); print('chunk', c, 'bad envs', bad.tolist()[:10])
    # replay this env step by step
    env.qpos[:] = q0; env.qvel[:] = v0; env.mocap_pos[:] = m0
    for t in range(T):
      o, r, d, info = env.step(acts[t])
      z = float(env.qpos[e, 11])
      if t % 10 == 0 or z < 0.012:
        print(t, 'peg', [round(float(x), 4) for x in env.qpos[e, 9:16]], 'hand', [round(float(x), 3) for x in o[e, :4]], 'v', [round(float(x), 2) for x in env.qvel[e, 9:12]])
      if z < -0.05: break
    break
