"""(round 2 tuning aid) Back-to-back multi-episode tabletop launches: time per launch, env-steps/s, algorithmic TB/s.  IMPL = debug switch (36: 8-step chunks,
38: episodes one after the other), E = episodes per launch; argument: ship | <tag of tools/ubench/libearl_<tag>.so>."""
import sys, time, torch
sys.path.insert(0,'/root/repo')
from earl_benchmark_amd import _abi
tag=sys.argv[1]
if tag!='ship': _abi.LIB_PATH=f'/root/repo/tools/ubench/libearl_{tag}.so'
import bench, os
lib=_abi.load(); lib.earl_debug_set_rollout_impl(int(os.environ.get('IMPL','0')))
n,T,E=4096,200,int(os.environ.get('E','28'))
env = bench.make_env(torch, n, T, 'sparse', 0, 'cuda')
acts = bench.synth_actions(torch, T, n, 0, 'cuda')
out = bench.alloc_out(torch, T, n, 'cuda', E)
for _ in range(3): env.rollout_episodes(acts, episodes=E, out=out)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): env.rollout_episodes(acts, episodes=E, out=out)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
print(tag, f'{dt*1e6:.1f} us per {E}-episode launch -> {n*T*E/dt/1e9:.1f} G env-steps/s, {n*(E*T*66+78)/dt/1e12:.2f} TB/s')
