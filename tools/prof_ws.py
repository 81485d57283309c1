"""Diagnostic: where the roles of the wave-specialised rollout kernel spend their cycles (instrumented variant 9)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import earl_benchmark_amd as eb
from earl_benchmark_amd import _abi
lib = _abi.load()
T = 200
for n in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['64', '4096', '262144'])]:
  L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
  _, env = L.get_envs()
  acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
  lib.earl_debug_set_rollout_impl(int(os.environ.get('PROF_IMPL', '9')))
  E = int(os.environ.get('PROF_EPISODES', '1'))      # > 1: the multi-episode instantiation (PROF_IMPL=29)
  for _ in range(5):
    if E > 1:
      env.rollout_episodes(acts, episodes=E)
    else:
      env.reset(); env.rollout(acts)
  torch.cuda.synchronize()
  buf = np.zeros(64 * 16, np.uint64)
  lib.earl_debug_read_ws_profile(buf.ctypes.data, buf.size)
  lib.earl_debug_set_rollout_impl(0)
  b = buf.reshape(64, 16)[:min(64, (n + 63) // 64)].astype(np.float64)
  m = np.median(b, axis=0)
  names = ['C.first_barrier', 'C.lds_read', 'C.compute', 'C.barrier', 'C.total', 'L.process+issue', 'L.barrier', 'L.total', 'S0.store', 'S0.barrier', 'S0.total', 'Slast.store', 'Slast.barrier', 'C1.barrier', 'Llast.process', 'Llast.barrier']
  print(f'n={n}: median cycles over {len(b)} workgroups (T={T}, K=8: 25 chunks)')
  for k, nm in enumerate(names):
    print(f'   {nm:18s} {m[k]:10.0f}  per step {m[k] / (200 * E):8.1f}')
