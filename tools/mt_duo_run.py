"""one minitaur rollout launch shape under rocprofv3: python3 tools/mt_duo_run.py <duo 0|1> [N] [T]   (the kernel's counters: tools/mt_duo_pmc.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.minitaur import Minitaur
mode = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
_abi.load().earl_debug_set_minitaur_duo(mode)
env = Minitaur(num_envs=n, seed=1234, scalar_api=False)
acts = (torch.rand(T, n, 8, generator=torch.Generator(device='cuda').manual_seed(99), device='cuda') * 2 - 1).float()
for _ in range(2):
  env.reset(); env.rollout(acts)
torch.cuda.synchronize()
