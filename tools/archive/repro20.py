"""(round 2 tuning aid) One 20-episode tabletop launch (the timed region of `bench.py --steps 20`) after various launch histories: a launch that is
the first to write its output rows runs slower, which is why bench.alloc_out zero-initialises."""
import sys, time, torch
sys.path.insert(0,'/root/repo')
import bench
import earl_benchmark_amd as eb
n, T = 4096, 200
env = bench.make_env(torch, n, T, 'sparse', 0, 'cuda')
acts = bench.synth_actions(torch, T, n, 0, 'cuda')
out = bench.alloc_out(torch, T, n, 'cuda', 25)
def launch(e):
  env.rollout_episodes(acts, episodes=e, out=tuple(t[:e] for t in out))
for seq in ([5,20],[25,20],[20,20],[5,20,20,20]):
  for e in seq[:-1]: launch(e)
  torch.cuda.synchronize()
  e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
  t0=time.perf_counter(); e0.record(); launch(seq[-1]); e1.record(); torch.cuda.synchronize(); w=time.perf_counter()-t0
  print(seq, f'kernel {e0.elapsed_time(e1)*1e3:.1f} us wall {w*1e6:.1f} us')
acts2 = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
launch(20); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout_episodes(acts2, episodes=20, out=tuple(t[:20] for t in out)); e1.record(); torch.cuda.synchronize()
print('uniform random actions instead of bench.synth_actions:', f'{e0.elapsed_time(e1)*1e3:.1f} us')
