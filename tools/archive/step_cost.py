import sys, time, torch
sys.path.insert(0,'/root/repo')
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
from earl_benchmark_amd.envs.kitchen import Kitchen
for cls, n, d in ((SawyerDoor, 8192, 4), (SawyerPeg, 8192, 4), (Kitchen, 2048, 9)):
  env = cls(num_envs=n, seed=1); env.reset()
  a = torch.rand(50, n, d, device='cuda')*2-1
  for t in range(5): env.step(a[t])
  torch.cuda.synchronize(); t0=time.perf_counter()
  for t in range(5,45): env.step(a[t])
  torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/40
  print(cls.__name__, n, 'step(): %.3f ms per call, %.2f M env-steps/s' % (dt*1e3, n/dt/1e6))
