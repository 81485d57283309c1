"""Per-phase cycle counts of one timestep of the articulated-body stepper (wave 0, lane 0; s_memtime-class counter).
Build (here or on the GPU box):  python tools/prof_physics.py --build     Run (GPU): python tools/prof_physics.py [N]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tools', 'ubench', 'libearl_physprof.so')
FLAGS = '--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -DEARL_PHYS_PROF'
NAMES = ['K1-2 joint + world transforms + C0 bounding tests', 'K3 subspace + inertia', 'C1-2 pair tests', 'K4 composite inertia', 'K5 mass matrix', 'K6-7 RNE + tau',
         'C3 contact rows', 'K8 weld / limit rows', 'K9b active-set Newton', 'K9a equality Hessian', '-', 'K10 Euler']


def build():
  srcs = [os.path.join(ROOT, 'earl_benchmark_amd', 'csrc', f) for f in ('physics.hip', 'physics_w8.hip', 'physics_mt.hip', 'physics_l64.hip', 'physics_kitchen.hip', 'tabletop.hip', 'glue.hip')]
  subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS.split() + ['-o', LIB] + srcs, check=True)


def rollout_counters(n, T, sel=None):
  """event counters of wave 0 over a random-action rollout of the Sawyer door (or, with --peg, peg) env (profiling build of the whole library)"""
  import torch
  sys.path.insert(0, ROOT)
  from earl_benchmark_amd import _abi
  _abi.LIB_PATH = LIB
  w8 = '--w8' in sys.argv                         # the door's eight-waves-per-CU build (its own translation unit and counters)
  reader = 'earl_debug_read_phys_profile_w8' if w8 else 'earl_debug_read_phys_profile'
  _abi.SIGNATURES[reader] = [C.c_void_p, C.c_int]
  _abi.SIGNATURES['earl_debug_read_wave_cycles_w8' if w8 else 'earl_debug_read_wave_cycles'] = [C.c_void_p]
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  _abi.SIGNATURES['earl_debug_set_prof_wave'] = [C.c_int, C.c_int]
  env = (SawyerPeg if '--peg' in sys.argv else SawyerDoor)(num_envs=n, seed=1234)
  lib = _abi.load()
  if w8:
    lib.earl_debug_set_door_variant(2)
    if sel is not None:
      lib.earl_debug_set_prof_wave_w8(C.c_int(sel // 8), C.c_int((sel % 8) * 64))      # (eight-wave workgroups)
  elif sel is not None:
    lib.earl_debug_set_prof_wave(sel // 4, (sel % 4) * 64)       # (four-wave workgroups: the peg build; the door's single-wave build has one wave per workgroup)
  read = getattr(lib, reader)
  out = (C.c_ulonglong * 32)()
  torch.manual_seed(0)
  acts = torch.rand(T, n, 4, device='cuda') * 2 - 1
  env.reset()
  torch.cuda.synchronize()
  read(out, 1)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  env.rollout(acts)
  e1.record()
  torch.cuda.synchronize()
  launch_ms = e0.elapsed_time(e1)
  read(out, 1)
  wc = (C.c_ulonglong * 4096)()
  getattr(lib, 'earl_debug_read_wave_cycles_w8' if w8 else 'earl_debug_read_wave_cycles')(wc)
  import numpy as np
  w = np.array(wc[:min(4096, n // 4)], dtype=np.float64) / (T * 5)
  print(f'  launch {launch_ms:.2f} ms; sum of all wave durations / (wave slots = min(waves, 1024 single-wave or 2048 eight-wave-build slots)) would be the balanced time')
  print(f'  wave durations, cycles per timestep: min {w.min():.0f}  p10 {np.percentile(w, 10):.0f}  median {np.median(w):.0f}  mean {w.mean():.0f}  p90 {np.percentile(w, 90):.0f}  '
        f'p99 {np.percentile(w, 99):.0f}  max {w.max():.0f}   (the launch lasts as long as its slowest wave)')
  ts = max(1, out[20])
  print(f'rollout N={n} T={T}: timesteps of wave {sel or 0}: {out[20]}; with a near block {out[21] / ts:.3f} (blocks per timestep {out[22] / ts:.2f}); '
        f'with contacts {out[23] / ts:.3f} (max contacts per env, mean {out[24] / ts:.2f}); Newton iterations per timestep {out[25] / ts:.2f}')
  print(f'  near blocks among the first six of the table (the door\'s 4-pair capsule blocks) per timestep: {out[31] / ts:.2f}')
  nc = out[26]
  print(f'  timesteps in which a contact joins the arm and the object (shared dense factorisation): {nc / ts:.3f}; active-set phase: {out[27] / max(1, nc):.0f} cycles in those, '
        f'{out[28] / max(1, ts - nc):.0f} in the others')
  if out[29]:                                      # (library built with -DEARL_PHYS_PROF_ALL)
    print(f'  all waves: {out[29]} wave-timesteps, coupled {out[30] / max(1, out[29]):.3f}')
  print(f'  active-set pass, cycles per timestep: edge weights {out[16] / ts:.0f}, Hessian columns {out[17] / ts:.0f}, factor + solve {out[18] / ts:.0f}, row test {out[19] / ts:.0f}')
  es = max(1, ts // 5)
  print(f'  env-step level, cycles per ENV step: action load + mocap {out[12] / es:.0f}, the 5 timesteps {out[13] / es:.0f}, guard + observation / reward {out[14] / es:.0f}, '
        f'state store + bookkeeping {out[15] / es:.0f}')
  tot = sum(out[:12])
  print(f'  cycles per timestep {tot / ts:.0f}: ' + ', '.join(f'{NAMES[i].split()[0]} {out[i] / ts:.0f}' for i in range(12) if NAMES[i] != '-'))
  return int(np.argmax(w))


def main():
  if '--build' in sys.argv:
    return build()
  if '--rollout' in sys.argv:
    n, T = 8192 if ('--w8' in sys.argv or '--full' in sys.argv) else 1024, 200 if '--peg' in sys.argv else 300
    slow = rollout_counters(n, T)
    if '--slowest' in sys.argv and ('--peg' in sys.argv or '--w8' in sys.argv):     # the same launch again, clocking the wave that took longest
      print(f'--- slowest wave: #{slow}')
      rollout_counters(n, T, sel=slow)
    return
  import numpy as np
  import torch
  sys.path.insert(0, ROOT)
  from earl_benchmark_amd import physics
  n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
  nsub = 5
  lib = C.CDLL(LIB)
  s, _ = physics.load_link_model('sawyer_door')
  buf = torch.from_numpy(np.frombuffer(bytes(s), dtype=np.uint8).copy()).cuda()
  rng = np.random.default_rng(0)
  qpos = rng.uniform(-0.3, 0.3, size=(n, 10)); qpos[:, 1] = -1.0; qpos[:, 7] = 0.02; qpos[:, 8] = -0.02; qpos[:, 9] = -1.0
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
  dq, dv = t(qpos), torch.zeros(n, 10, dtype=torch.float64, device='cuda')
  mp = t(np.tile([0.0, 0.6, 0.2], (n, 1))); mq = t(np.tile([1.0, 0, 1, 0], (n, 1))); ctrl = t(rng.uniform(-1, 1, size=(n, 2)))
  col = physics.load_collision_model(_) if '--nocol' not in sys.argv else None
  cbuf = torch.from_numpy(np.frombuffer(bytes(col), dtype=np.uint8).copy()).cuda() if col is not None else None
  args = [C.c_void_p(buf.data_ptr()), C.c_void_p(cbuf.data_ptr() if cbuf is not None else None), C.c_int32(10), C.c_int32(n), C.c_int32(nsub)] + [C.c_void_p(x.data_ptr()) for x in (dq, dv, mp, mq, ctrl)] + [None, None]
  out = (C.c_ulonglong * 32)()
  for rep in range(3):
    lib.earl_physics_step(*args)
    torch.cuda.synchronize()
    lib.earl_debug_read_phys_profile(out, 1)
  tot = sum(out[:12])
  print(f'N={n}: cycles per timestep of wave 0 (= 4 envs at 16 lanes per env), total {tot / nsub:.0f}')
  for i, nm in enumerate(NAMES):
    if nm != '-':
      print(f'  {nm:32s} {out[i] / nsub:9.0f}  {100.0 * out[i] / tot:5.1f} %')


if __name__ == '__main__':
  main()
