import sys, time, torch
sys.path.insert(0,'/root/repo')
from earl_benchmark_amd import _abi
tag=sys.argv[1]
if tag!='ship': _abi.LIB_PATH=f'/root/repo/tools/ubench/libearl_{tag}.so'
import earl_benchmark_amd as eb
lib=_abi.load()
n,E,T=4096,16,208
acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
_, env = L.get_envs()
for impl in (0,):
  lib.earl_debug_set_rollout_impl(impl)
  f = lambda: env.rollout_episodes(acts, episodes=E)
  for _ in range(3): f()
  torch.cuda.synchronize(); t0=time.perf_counter()
  for _ in range(20): f()
  torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
  print(tag,'impl',impl,f'{dt/E/T*1e9:.1f} ns per step')
