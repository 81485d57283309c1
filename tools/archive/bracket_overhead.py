import sys, time, torch
sys.path.insert(0,'/root/repo')
import bench
n, T = 4096, 200
env = bench.make_env(torch, n, T, 'sparse', 0, 'cuda')
acts = bench.synth_actions(torch, T, n, 0, 'cuda')
out = bench.alloc_out(torch, T, n, 'cuda', 28)
views = tuple(t[:20] for t in out)
for _ in range(3): env.rollout_episodes(acts, episodes=20, out=views)
torch.cuda.synchronize()
ts=[]
for _ in range(20):
  torch.cuda.synchronize()
  t0=time.perf_counter(); env.rollout_episodes(acts, episodes=20, out=views); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
  ts.append((t1-t0, t2-t0))
ts.sort(key=lambda x:x[1])
print('python call (async) median %.1f us; call + sync median %.1f us' % (sorted(t[0] for t in ts)[10]*1e6, ts[10][1]*1e6))
t0=time.perf_counter()
for _ in range(100): torch.cuda.synchronize()
print('empty synchronize %.1f us' % ((time.perf_counter()-t0)/100*1e6))
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
t0=time.perf_counter()
for _ in range(100): e0.record()
torch.cuda.synchronize()
print('event record %.1f us' % ((time.perf_counter()-t0)/100*1e6))
