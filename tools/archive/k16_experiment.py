"""(round 2 tuning aid) Multi-episode tabletop launch: per-episode launches of the plain kernel (impl 1, the reference), 8-step chunks (impl 36) and the shipped
16-step chunks (impl 0) on the same seeds: time per step and bit-identity of all outputs.  python tools/k16_experiment.py [T]"""
import sys, time, torch
sys.path.insert(0,'/root/repo')
import earl_benchmark_amd as eb
from earl_benchmark_amd import _abi
lib=_abi.load()
n,E,T=4096,32,200
T=int(sys.argv[1]) if len(sys.argv)>1 else 208
acts = (torch.rand(T, n, 3, device='cuda') * 2 - 1).contiguous()
res={}
for impl in (1, 36, 0, 38):        # 38: the episodes of a launch one after the other        # 1 = the plain kernel, one launch per episode: the reference
  L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)     # fresh env: same Philox counters in both runs
  _, env = L.get_envs()
  lib.earl_debug_set_rollout_impl(impl)
  f = lambda: env.rollout_episodes(acts, episodes=E)
  for _ in range(3): out=f()
  torch.cuda.synchronize(); t0=time.perf_counter()
  for _ in range(20): out=f()
  torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
  res[impl]=[v.clone() for v in (out if isinstance(out,(tuple,list)) else out.values()) if torch.is_tensor(v)]
  print('impl',impl,'T',T,f'{dt*1e6:.1f} us -> {dt/E/T*1e9:.1f} ns per step')
lib.earl_debug_set_rollout_impl(0)
print('identical to the per-episode launches: 8-step chunks', all(torch.equal(a,b) for a,b in zip(res[1],res[0])), ' 16-step chunks', all(torch.equal(a,b) for a,b in zip(res[1],res[0])))
for i,(a,b) in enumerate(zip(res[1],res[0])):
  if not torch.equal(a,b):
    d=(a!=b)
    idx=d.nonzero()
    print('tensor',i,a.shape,a.dtype,'mismatches',int(d.sum()),'first',idx[:5].tolist(),'episodes with diffs',sorted(set(idx[:,0].tolist()))[:10], 'steps', sorted(set(idx[:,1].tolist()))[:20])
