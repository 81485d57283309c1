"""(round 2 tuning aid) Does a launch that follows an idle gap run slower?  20-episode tabletop launches at N = 4096 after gaps of various lengths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import earl_benchmark_amd as eb
n, T, E = 4096, 200, 20
L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
_, env = L.get_envs()
acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
out = env.rollout_episodes(acts, episodes=E)
torch.cuda.synchronize()
for gap in (0.0, 50e-6, 200e-6, 1e-3, 10e-3, 100e-3):
  ks, ws = [], []
  for rep in range(8):
    env.rollout_episodes(acts, episodes=E, out=out) if 'out' in env.rollout_episodes.__code__.co_varnames else env.rollout_episodes(acts, episodes=E)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + gap
    while time.perf_counter() < t_end: pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    env.rollout_episodes(acts, episodes=E)
    e1.record()
    torch.cuda.synchronize()
    ws.append(time.perf_counter() - t0); ks.append(e0.elapsed_time(e1))
  print(f'gap {gap*1e6:8.0f} us: kernel (events) median {sorted(ks)[4]*1e3:.1f} us, wall median {sorted(ws)[4]*1e6:.1f} us')
