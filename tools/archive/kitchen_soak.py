import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np, time
from earl_benchmark_amd.envs.kitchen import Kitchen
n,T=2048,400
env=Kitchen(num_envs=n, seed=3); env.reset()
g=torch.Generator(device='cuda').manual_seed(0)
lo=torch.tensor(env.model.tables['jnt_range'][:,0],device='cuda'); hi=torch.tensor(env.model.tables['jnt_range'][:,1],device='cuda')
worst=torch.zeros(23,dtype=torch.float64,device='cuda'); moved=torch.zeros(14,dtype=torch.float64,device='cuda'); q0=env.qpos[:,9:].clone()
torch.cuda.synchronize(); t0=time.time()
for t in range(T):
  a=torch.rand(n,9,generator=g,device='cuda')*2-1
  o,r,d,info=env.step(a)
  worst=torch.maximum(worst, torch.maximum(lo-env.qpos, env.qpos-hi).amax(0))
  moved=torch.maximum(moved,(env.qpos[:,9:]-q0).abs().amax(0))
torch.cuda.synchronize(); print('time',time.time()-t0,'env-steps/s',n*T/(time.time()-t0))
print('limit violation per joint', worst.cpu().numpy().round(4))
print('fixture movement', moved.cpu().numpy().round(4))
print('qvel max', env.qvel.abs().amax(0).cpu().numpy().round(2), 'fails', int(env.fail_count.sum()))
