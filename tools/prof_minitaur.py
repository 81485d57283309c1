"""Per-phase cycle counts of the minitaur's timestep (wave 0 of workgroup 0: two envs): profiling build of csrc/physics_mt.hip (-DEARL_PHYS_PROF).
Build (here or on the GPU box):  python tools/prof_minitaur.py --build     Run (GPU): python tools/prof_minitaur.py [N] [T]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'earl_benchmark_amd', 'csrc')
LIB = os.path.join(ROOT, 'tools', 'ubench', 'libearl_mtprof.so')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC'
NAMES = ['K1-2 joint + world transforms + C0 bounding tests', 'K3 subspace + inertia', 'C1-2 pair tests', 'K4 composite inertia', 'K5 mass matrix', 'K6-7 RNE + tau',
         'C3 contact rows', 'K8 weld / connect / limit rows', 'K9b active-set Newton', 'K9a equality Hessian', '-', 'K10 Euler']

if '--build' in sys.argv:
  obj = os.path.join(ROOT, 'tools', 'ubench', 'physics_mt_prof.o')
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS.split() + ['-DEARL_PHYS_PROF', '-c', '-o', obj, os.path.join(CSRC, 'physics_mt.hip')], check=True)
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS.split() + ['-shared', '-o', LIB, obj] + [os.path.join(CSRC, f) for f in ('tabletop.o', 'glue.o', 'physics.o', 'physics_w8.o', 'physics_l64.o', 'physics_kitchen.o')], check=True)
  os.remove(obj)
  sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
_abi.LIB_PATH = LIB
_abi.SIGNATURES['earl_debug_read_phys_profile_mt'] = [C.c_void_p, C.c_int]
_abi.SIGNATURES['earl_debug_read_wave_cycles_mt'] = [C.c_void_p]
from earl_benchmark_amd.envs.minitaur import Minitaur
args = [int(x) for x in sys.argv[1:] if x.isdigit()]
n, T = (args + [4096, 100])[:2] if len(args) < 2 else args[:2]
_abi.SIGNATURES['earl_debug_set_prof_wave_mt'] = [C.c_int, C.c_int]
env = Minitaur(num_envs=n, seed=1234, scalar_api=False)
lib = _abi.load()
if os.environ.get('PROF_DUO'):          # the two-waves-per-SIMD kernel: PROF_THREAD = 0 clocks a first-half wave, 256 its partner
  lib.earl_debug_set_minitaur_duo(1)
  lib.earl_debug_set_prof_wave_mt(int(os.environ.get('PROF_BLOCK', '0')), int(os.environ.get('PROF_THREAD', '0')))
out = (C.c_ulonglong * 32)()
torch.manual_seed(0)
acts = torch.rand(T, n, 8, device='cuda') * 2 - 1
torch.cuda.synchronize()
lib.earl_debug_read_phys_profile_mt(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
env.rollout(acts)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
lib.earl_debug_read_phys_profile_mt(out, 1)
import numpy as np
wc = (C.c_ulonglong * 4096)()
lib.earl_debug_read_wave_cycles_mt(wc)
w = np.array(wc[:min(4096, (n + 1) // 2)], dtype=np.float64) / (T * 5)
if os.environ.get('PROF_DUO'):
  nw = min(2048, (n + 15) // 16 * 8)
  tot, wait = np.array(wc[:nw], dtype=np.float64).reshape(-1, 2, 4) / (T * 5 * 2), np.array(wc[2048:2048 + nw], dtype=np.float64).reshape(-1, 2, 4) / (T * 5 * 2)     # [workgroup, role, pair]; per slot
  work = tot - wait
  pct = lambda x: f'min {x.min():.0f}  p10 {np.percentile(x, 10):.0f}  median {np.median(x):.0f}  mean {x.mean():.0f}  p90 {np.percentile(x, 90):.0f}  p99 {np.percentile(x, 99):.0f}  max {x.max():.0f}'
  print('  two-wave kernel, cycles per slot over all wave pairs: duration ' + pct(tot[:, 0]))
  print('    first-half waves, own work ' + pct(work[:, 0]) + '\n    second-half waves, own work ' + pct(work[:, 1]))
  k = int(tot[:, 0].reshape(-1).argmax())
  print(f'    slowest pair: workgroup {k // 4} pair {k % 4} (PROF_BLOCK={k // 4} PROF_THREAD={(k % 4) * 64} / {256 + (k % 4) * 64}): first half {work[k // 4, 0, k % 4]:.0f}, second half {work[k // 4, 1, k % 4]:.0f}')
  w = tot[:, 0].reshape(-1) * 2
print(f'  wave durations, cycles per timestep: min {w.min():.0f}  p10 {np.percentile(w, 10):.0f}  median {np.median(w):.0f}  mean {w.mean():.0f}  p90 {np.percentile(w, 90):.0f}  p99 {np.percentile(w, 99):.0f}  max {w.max():.0f}')
ts = max(1, out[20])
print(f'minitaur N={n} T={T}: launch {ms:.1f} ms = {n * T / ms / 1e3:.2f} M env-steps/s; wave 0: {out[20]} timesteps, with contacts {out[23] / ts:.3f} '
      f'(max contacts per env, mean {out[24] / ts:.2f}), Newton iterations per timestep {out[25] / ts:.2f}')
print(f'  timesteps with more than three passes {out[10] / ts:.3f}, with all eight {out[12] / ts:.4f}')
print(f'  active-set pass, cycles per timestep: edge weights {out[16] / ts:.0f}, Hessian columns {out[17] / ts:.0f}, factor + solve {out[18] / ts:.0f}, row test {out[19] / ts:.0f}')
if out[28]:
  print(f'  two-waves-per-SIMD kernel, this wave: {out[28]} slots, work {out[26] / out[28]:.0f} cycles per slot, barrier wait {out[27] / out[28]:.0f}')
tot = sum(out[i] for i in range(12) if NAMES[i] != '-')   # slot 10 holds a pass-count tally, not cycles
print(f'  cycles per timestep {tot / ts:.0f}: ' + ', '.join(f'{NAMES[i].split()[0]} {out[i] / ts:.0f}' for i in range(12) if NAMES[i] != '-'))
