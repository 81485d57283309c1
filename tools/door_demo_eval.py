import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); 
import numpy as np, torch
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
DEMOS=os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/earl_benchmark_amd/demonstrations/sawyer_door'
eps = []
for d in ('forward', 'reverse'):
  z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
  ends = np.nonzero(z['terminals'].ravel())[0] + 1
  for s0, e0 in zip([0] + list(ends[:-1]), ends):
    eps.append((d, z['observations'][s0], z['actions'][s0:e0], z['next_observations'][s0:e0]))
n, T = len(eps), max(len(e[2]) for e in eps)
env = SawyerDoor(num_envs=n); env.reset()
want_handle = np.stack([e[1][4:7] for e in eps]).astype(np.float64)
best, err = np.zeros(n), np.full(n, 1e9)
for a in np.linspace(-1.5, 0.1, 801):
  env.qpos[:, 9] = a
  e2 = ((env._get_obs()[:, 4:7].cpu().numpy() - want_handle) ** 2).sum(1)
  m = e2 < err; best[m], err[m] = a, e2[m]
env.qpos[:, 9] = torch.from_numpy(best).cuda()
env.goal_t[:] = torch.from_numpy(np.stack([e[1][7:] for e in eps]).astype(np.float64)).cuda()
acts = np.zeros((T, n, 4), np.float32)
for i, e in enumerate(eps): acts[:len(e[2]), i] = e[2]
out = env.rollout(torch.from_numpy(acts).cuda())
obs, suc = out['obs'].cpu().numpy(), out['success'].cpu().numpy()
for i, e in enumerate(eps):
  L = len(e[2]); o, w = obs[:L, i], e[3]
  eh = np.linalg.norm(o[:, 4:7] - w[:, 4:7], axis=1)
  moved = np.linalg.norm(w[:, 4:7] - w[0, 4:7], axis=1)
  t0 = int(np.argmax(moved > 0.002)); bad = np.nonzero(eh[t0:] > 0.005)[0]; keep = L - t0 if len(bad) == 0 else int(bad[0])
  start = np.linalg.norm(w[0, 4:7] - w[0, 11:14]); closest = np.linalg.norm(o[:, 4:7] - o[:, 11:14], axis=1).min()
  print(e[0], i, 'L', L, 't0', t0, 'steps<5mm', keep, 'travel %.3f' % moved[min(t0 + keep, L - 1)], 'rms %.3f' % np.sqrt((eh**2).mean()), 'final %.3f' % eh[-1],
        'success', bool(suc[:L, i].any()), 'closest %.3f start %.3f' % (closest, start), 'grip err max %.3f' % np.abs(o[:, 3] - w[:, 3]).max(), 'hand err max %.3f' % np.linalg.norm(o[:, :3] - w[:, :3], axis=1).max())
