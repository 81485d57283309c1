"""(round 2 tuning aid) Per-step vs per-episode cost of the multi-episode tabletop launch: time E episodes of T steps for several T (N = 4096)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import earl_benchmark_amd as eb
n, E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 16
for T in (200, 400, 800):
  L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
  _, env = L.get_envs()
  acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
  res = {}
  for mode, ne in (('one', 1), ('multi', E)):
    f = (lambda: env.rollout_episodes(acts, episodes=ne)) if ne > 1 else (lambda: (env.reset(), env.rollout(acts)))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / 20
  per_ep = (res['multi'] - res['one']) / (E - 1)
  print(f'T={T}: one episode {res["one"]*1e6:.1f} us (wall), {E} episodes {res["multi"]*1e6:.1f} us -> {per_ep*1e6:.2f} us per further episode = {per_ep/T*1e9:.1f} ns per step')
