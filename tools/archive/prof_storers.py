import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import earl_benchmark_amd as eb
from earl_benchmark_amd import _abi
lib = _abi.load()
T, n = 200, 2048
for E in (1, 16):
  L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
  _, env = L.get_envs()
  acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
  lib.earl_debug_set_rollout_impl(29)
  for _ in range(3):
    if E > 1: env.rollout_episodes(acts, episodes=E)
    else: env.reset(); env.rollout(acts)
  torch.cuda.synchronize()
  buf = np.zeros(64 * 16, np.uint64)
  lib.earl_debug_read_ws_profile(buf.ctypes.data, buf.size)
  lib.earl_debug_set_rollout_impl(0)
  b = buf.reshape(64, 16).astype(np.float64)
  print('E', E, 'storers of workgroup 0: (store, barrier) cycles per step:', [(round(b[32+s,8]/(T*E),1), round(b[32+s,9]/(T*E),1)) for s in range(8)])
