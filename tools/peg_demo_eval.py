"""Open-loop replay of the 20 sawyer_peg demonstrations on the GPU env; prints the per-episode figures DESIGN.md section 10 quotes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
DEMOS = ROOT + '/earl_benchmark_amd/demonstrations/sawyer_peg'
for d in ('forward', 'reverse'):
  z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
  ends = np.nonzero(z['terminals'].ravel())[0] + 1
  eps = [(z['observations'][s], z['actions'][s:e], z['next_observations'][s:e]) for s, e in zip([0] + list(ends[:-1]), ends)]
  n, T = len(eps), max(len(e[1]) for e in eps)
  env = SawyerPeg(num_envs=n); o0 = env.reset()
  if d == 'forward':
    print('reset obs', o0[0, :7].cpu().numpy().round(5), 'recorded', eps[0][0][:7].round(5))
  heads = np.stack([e[0][4:7] for e in eps]).astype(np.float64)
  env.qpos[:, 9:12] = torch.from_numpy(heads + np.array([0.1, 0.0, 0.0])).cuda(); env.qvel[:, 9:] = 0
  env.goal_t[:] = torch.from_numpy(np.stack([e[0][7:] for e in eps]).astype(np.float64)).cuda()
  acts = np.zeros((T, n, 4), np.float32)
  for i, e in enumerate(eps): acts[:len(e[1]), i] = e[1]
  out = env.rollout(torch.from_numpy(acts).cuda())
  obs, suc = out['obs'].cpu().numpy(), out['success'].cpu().numpy()
  for i, e in enumerate(eps):
    L = len(e[1]); o, w = obs[:L, i], e[2]
    eh = np.linalg.norm(o[:, :3] - w[:, :3], axis=1); ep = np.linalg.norm(o[:, 4:7] - w[:, 4:7], axis=1)
    bad = np.nonzero(ep > 0.005)[0]
    print(d, i, 'L', L, 'hand rms %.4f max %.4f' % (np.sqrt((eh**2).mean()), eh.max()), 'peg rms %.4f max %.4f final %.4f' % (np.sqrt((ep**2).mean()), ep.max(), ep[-1]),
          'peg<5mm until', L if len(bad) == 0 else int(bad[0]), 'lift %.3f/%.3f' % (o[:, 6].max(), w[:, 6].max()), 'success', bool(suc[L - 1, i]), 'grip err max %.3f' % np.abs(o[:, 3] - w[:, 3]).max())
