"""Per-phase cycle counts of the kitchen stepper inside the fused rollout (profiling build of tools/prof_physics.py --build): wave durations of the launch,
then the phases of wave 0 and of the SLOWEST wave (the launch lasts as long as it).   python tools/prof_kitchen_phases.py [N] [T]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from earl_benchmark_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', 'libearl_physprof.so')
_abi.SIGNATURES['earl_debug_read_phys_profile_kitchen'] = [C.c_void_p, C.c_int]      # (the kitchen's translation unit, csrc/physics_kitchen.hip, has its own counters)
_abi.SIGNATURES['earl_debug_read_wave_cycles_kitchen'] = [C.c_void_p]
_abi.SIGNATURES['earl_debug_set_prof_wave_kitchen'] = [C.c_int, C.c_int]
from earl_benchmark_amd.envs.kitchen import Kitchen
NAMES = ['K1-2', 'K3', 'C1-2', 'K4', 'K5', 'K6-7', 'C3', 'K8', 'K9b', 'K9a', 'wait X', 'K10', 'wait Y', '-', '-', '-']
DUO = '--duo' in sys.argv                               # four waves per env (n <= CUs): the phases of every wave of a workgroup, incl. their waits at the timestep's barriers
nums = [int(x) for x in sys.argv[1:] if x.isdigit()]
n, T = (nums + [2048, 100])[:2] if len(nums) < 2 else nums[:2]
lib = _abi.load()
out = (C.c_ulonglong * 32)()
g = torch.Generator(device='cuda').manual_seed(0)
acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1


def run(block, thread):
  env = Kitchen(num_envs=n, seed=3); env.reset()
  lib.earl_debug_set_solo(3 if DUO else 0)               # (two envs per wave whatever the batch size: the wave indexing below assumes it)
  lib.earl_debug_set_prof_wave_kitchen(block, thread)
  torch.cuda.synchronize()
  lib.earl_debug_read_phys_profile_kitchen(out, 1)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); env.rollout(acts); e1.record(); torch.cuda.synchronize()
  lib.earl_debug_read_phys_profile_kitchen(out, 1)
  wc = (C.c_ulonglong * 4096)()
  lib.earl_debug_read_wave_cycles_kitchen(wc)
  return e0.elapsed_time(e1), np.array(wc[:(n + 1) // 2], dtype=np.float64) / (T * 40), list(out)


def show(tag, o):
  ts = max(1, o[20]) if not DUO else T * 40; tot = sum(o[:13])
  print(f'{tag}: timesteps {o[20]}; near block {o[21] / ts:.3f}; with contacts {o[23] / ts:.3f} (max per env, mean {o[24] / ts:.2f}); passes per timestep {o[25] / ts:.2f}; coupled {o[26] / ts:.3f}')
  print(f'  cycles per timestep {tot / ts:.0f}: ' + ', '.join(f'{NAMES[i]} {o[i] / ts:.0f}' for i in range(16) if NAMES[i] != '-' and (DUO or i not in (10, 12))))
  print(f'  active-set pass: edge weights {o[16] / ts:.0f}, Hessian columns {o[17] / ts:.0f}, factor + solve {o[18] / ts:.0f}, row test {o[19] / ts:.0f}; '
        f'K9b in coupled timesteps {o[27] / max(1, o[26]):.0f}, in the others {o[28] / max(1, ts - o[26]):.0f}; per coupled timestep: fixtures inverse {o[13] / max(1, o[26]):.0f}, Schur rows {o[14] / max(1, o[26]):.0f}, arm solve {o[15] / max(1, o[26]):.0f}')


if DUO:
  n = min(n, 256)
  acts = acts[:, :n].contiguous()
  for blk in (0, 100):
    ms, w, o = run(blk, 0)
    print(f'kitchen rollout N={n} T={T}, four waves per env: launch {ms:.1f} ms = {ms * 1e-3 * 2.4e9 / (T * 40):.0f} cycles per timestep at 2.4 GHz')
    show(f'workgroup {blk} wave 0 = B (constraint rows | contact rows, active set, integration)', o)
    for wv, what in ((1, 'A (mass matrix | equality Hessian)'), (2, 'bias forces'), (3, 'bounding tests + collision')):
      ms, w, o = run(blk, 64 * wv)
      show(f'workgroup {blk} wave {wv} = {what}; "wait X" / "wait Y" = its time in the two barriers of a timestep', o)
  sys.exit(0)
ms, w, o = run(0, 0)
print(f'kitchen rollout N={n} T={T}: launch {ms:.1f} ms = {n * T / ms / 1e3:.3f} M env-steps/s')
print(f'  wave durations, cycles per timestep: min {w.min():.0f}  p10 {np.percentile(w, 10):.0f}  median {np.median(w):.0f}  mean {w.mean():.0f}  p90 {np.percentile(w, 90):.0f}  '
      f'p99 {np.percentile(w, 99):.0f}  max {w.max():.0f}   (the launch lasts as long as its slowest wave)')
show('wave 0', o)
slow = int(np.argmax(w))
ms2, w2, o2 = run(slow // 4, (slow % 4) * 64)
show(f'slowest wave (#{slow}, {w2[slow]:.0f} cycles per timestep)', o2)
