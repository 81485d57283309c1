"""Per-phase cycle counts of the kitchen stepper inside the fused rollout (wave 0 of workgroup 0; profiling build of tools/prof_physics.py --build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', 'libearl_physprof.so')
_abi.SIGNATURES['earl_debug_read_phys_profile'] = [C.c_void_p, C.c_int]
from earl_benchmark_amd.envs.kitchen import Kitchen
NAMES = ['K1-2', 'K3', 'C1-2', 'K4', 'K5', 'K6-7', 'C3', 'K8', 'K9b', 'K9a', '-', 'K10']
n, T = 2048, 100
env = Kitchen(num_envs=n, seed=3); env.reset()
lib = _abi.load()
out = (C.c_ulonglong * 32)()
g = torch.Generator(device='cuda').manual_seed(0)
acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1
env.rollout(acts[:5]); torch.cuda.synchronize()
lib.earl_debug_read_phys_profile(out, 1)
env.rollout(acts); torch.cuda.synchronize()
lib.earl_debug_read_phys_profile(out, 1)
ts = max(1, out[20])
tot = sum(out[:12])
print(f'kitchen rollout N={n} T={T}: timesteps of wave 0: {out[20]}; with contacts {out[23] / ts:.3f}; Newton passes per timestep {out[25] / ts:.2f}; coupled {out[26] / ts:.3f}')
print(f'  cycles per timestep {tot / ts:.0f}: ' + ', '.join(f'{NAMES[i]} {out[i] / ts:.0f}' for i in range(12) if NAMES[i] != '-'))
print(f'  active-set pass: edge weights {out[16] / ts:.0f}, Hessian columns {out[17] / ts:.0f}, factor + solve {out[18] / ts:.0f}, row test {out[19] / ts:.0f}')
