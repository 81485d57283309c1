"""Times the rollout kernel variants selectable through earl_debug_set_rollout_impl (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import earl_benchmark_amd as eb
from earl_benchmark_amd import _abi

lib = _abi.load()
impls = [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,1,2,3,4,5,6,7'.split(','))]
sizes = [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '64,4096,65536,262144'.split(','))]
T = int(os.environ.get('TUNE_T', '200'))
for n in sizes:
  L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
  _, env = L.get_envs()
  acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
  out = (torch.empty(T, n, 12, device='cuda'), torch.empty(T, n, device='cuda'), torch.empty(T, n, dtype=torch.bool, device='cuda'), torch.empty(T, n, dtype=torch.bool, device='cuda'))
  for impl in impls:
    lib.earl_debug_set_rollout_impl(impl)
    reps = max(5, min(200, int(3e9 // (n * T * 66))))
    for _ in range(3):
      env.rollout(acts, out=out, reset_first=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in ev:
      a.record(); env.rollout(acts, out=out, reset_first=True); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    med = ms[len(ms) // 2]
    print(f'n={n:8d} impl={impl} kernel_us={med * 1e3:9.1f}  GB/s={n * T * 66 / (med * 1e-3) / 1e9:8.1f}  Gsteps/s={n * T / (med * 1e-3) / 1e9:7.2f}', flush=True)
  lib.earl_debug_set_rollout_impl(0)
  del env, acts, out
  torch.cuda.empty_cache()
