"""Time the minitaur bench step (reset + T-step fused rollout, N envs) through alternative builds of csrc/physics_mt.hip:
   python tools/bench_mt_variant.py --build <tag> [-DEARL_MT_BLOCKS=1 ...]   (here or on the GPU box: tools/ubench/libearl_mt_<tag>.so)
   python tools/bench_mt_variant.py <tag>|ship [N] [T] [generic]             (GPU)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'earl_benchmark_amd', 'csrc')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC'
if sys.argv[1] == '--build':
  tag, defs = sys.argv[2], sys.argv[3:]
  obj = os.path.join(ROOT, 'tools', 'ubench', f'physics_mt_{tag}.o')
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS.split() + defs + ['-c', '-o', obj, os.path.join(CSRC, 'physics_mt.hip')], check=True)
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS.split() + ['-shared', '-o', os.path.join(ROOT, 'tools', 'ubench', f'libearl_mt_{tag}.so'), obj] +
                 [os.path.join(CSRC, f) for f in ('tabletop.o', 'glue.o', 'physics.o', 'physics_w8.o', 'physics_l64.o', 'physics_kitchen.o')], check=True)
  os.remove(obj)
  sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
tag = sys.argv[1]
if tag != 'ship':
  _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_mt_{tag}.so')
from earl_benchmark_amd.envs.minitaur import Minitaur
nums = [int(x) for x in sys.argv[2:] if x.isdigit()]
n, T = (nums + [4096, 1000])[:2] if len(nums) < 2 else nums[:2]
if 'generic' in sys.argv:
  _abi.load().earl_debug_set_minitaur_stepper(0)
env = Minitaur(num_envs=n, seed=1234, scalar_api=False)
g = torch.Generator(device='cuda').manual_seed(99)
acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1).float()
env.reset(); r = env.rollout(acts)
torch.cuda.synchronize()
reps = 2
t0 = time.perf_counter()
for _ in range(reps):
  env.reset(); r = env.rollout(acts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f'{tag:10s} minitaur N={n} T={T}{" generic stepper" if "generic" in sys.argv else ""}: {dt * 1e3:8.2f} ms per reset + rollout, {n * T / dt / 1e6:7.2f} M env-steps/s, '
      f'obs checksum {float(r["obs"].sum()):.9e}, failed env steps {int(env.fail_count.sum())}')
