"""Host-side cost of the Python call path (GPU box): tiny launches so the GPU is never the limiter."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import earl_benchmark_amd as eb
n, T = 64, 2
L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
_, env = L.get_envs()
acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
out = (torch.empty(T, n, 12, device='cuda'), torch.empty(T, n, device='cuda'), torch.empty(T, n, dtype=torch.bool, device='cuda'), torch.empty(T, n, dtype=torch.bool, device='cuda'))
def timeit(name, fn, reps=3000):
  for _ in range(200): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
  print(f'{name:40s} host {1e6 * (t1 - t0) / reps:7.2f} us/call   (drain {1e6 * (t2 - t1) / reps:6.2f})')
timeit('rollout(out=, reset_first=True)', lambda: env.rollout(acts, out=out, reset_first=True))
timeit('rollout(out=)', lambda: env.rollout(acts, out=out))
timeit('rollout()', lambda: env.rollout(acts))
timeit('step()', lambda: env.step(acts[0]))
timeit('reset()', lambda: env.reset())
ev = torch.cuda.Event(enable_timing=True)
timeit('event.record()', lambda: ev.record())
timeit('current_stream().cuda_stream', lambda: torch.cuda.current_stream(env.unwrapped.device).cuda_stream)
timeit('with torch.cuda.device', lambda: torch.cuda.device(env.unwrapped.device).__enter__())
timeit('torch.empty x4', lambda: (torch.empty(n, 12, device='cuda'), torch.empty(n, device='cuda'), torch.empty(n, dtype=torch.bool, device='cuda'), torch.empty(n, dtype=torch.bool, device='cuda')))
