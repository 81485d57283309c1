import torch, sys
sys.path.insert(0,'/root/repo')
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.kitchen import Kitchen
lib=_abi.load()
n,T=203,200
g=torch.Generator(device='cuda').manual_seed(9)
acts=torch.rand(T,n,9,generator=g,device='cuda')*2-1
acts[:, :40, 2] -= 0.6
import os
if os.environ.get('NAN','1')=='1': acts[3,7]=float('nan')
res={}
for mode in (0,2,3):
    lib.earl_debug_set_solo(mode)
    env=Kitchen(num_envs=n, seed=21); env.reset()
    out=env.rollout(acts)
    res[mode]=(torch.nan_to_num(out['obs'].clone(), nan=123.0), env.qpos.clone())
lib.earl_debug_set_solo(-1)
for mode in (2,3):
    d=(res[mode][0]-res[0][0]).abs()
    bad=(d>0).nonzero()
    print('mode',mode,'max diff',float(d.max()),'first differing (t, env, k):', bad[0].tolist() if len(bad) else None, 'n differing envs', len(set(bad[:,1].tolist())) if len(bad) else 0)
    if len(bad):
        t0=int(bad[:,0].min()); print(' first t',t0, 'envs at first t', sorted(set(bad[bad[:,0]==t0][:,1].tolist()))[:10], 'ks', sorted(set(bad[bad[:,0]==t0][:,2].tolist()))[:12])
