"""GPU box: time earl_tabletop_eval_episodes (4096 envs x 200 steps x E episodes, every episode with its OWN actions) for the tuning
variants of the multi-episode kernel (earl_debug_set_rollout_impl 0 / 36 / 40-46) and 1-2 workgroups per CU; optionally the role
stamps of the instrumented build (impl 29).  usage: own_actions_experiment.py [impls] [wgs_per_cu list] [E] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import earl_benchmark_amd as eb
from earl_benchmark_amd import _abi
if os.environ.get('EARL_WS_TAG'):
  _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f"libearl_ws_{os.environ['EARL_WS_TAG']}.so")
lib = _abi.load()
impls = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,36,40,41,42,43,44,45,46').split(',')]
occs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1,2').split(',')]
E = int(sys.argv[3]) if len(sys.argv) > 3 else 28
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
T = 200
L = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=n, eval_horizon=T, scalar_api=False)
_, env = L.get_envs()
Emax = E * 2
ROT = int(os.environ.get('ROTATE', '1'))      # distinct action tensors used round-robin by the timed launches (> 256 MB Infinity Cache in total)
acts_r = [(torch.rand(Emax, T, n, 3, device='cuda') * 2 - 1).contiguous() for _ in range(ROT)]
acts = acts_r[0]
out = (torch.zeros(Emax, T, n, 12, device='cuda'), torch.zeros(Emax, T, n, device='cuda'), torch.zeros(Emax, T, n, dtype=torch.bool, device='cuda'),
       torch.zeros(Emax, T, n, dtype=torch.bool, device='cuda'))
ref = None
for occ in occs:
  lib.earl_debug_set_rollout_wgs_per_cu(occ)
  for own in ((True, False) if not os.environ.get('OWN_ONLY') else (True,)):
    for impl in impls:
      Ex = E * occ                       # the same number of rounds per group
      a = acts[:Ex] if own else acts[0]
      o = tuple(t[:Ex] for t in out)
      lib.earl_debug_set_rollout_impl(impl)
      try:
        for _ in range(3):
          env.rollout_episodes(a, episodes=Ex, out=o)
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for r_ in range(reps):
          env.rollout_episodes(acts_r[r_ % ROT][:Ex] if own else a, episodes=Ex, out=o)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        chk = float(o[0][:E].double().sum()) if own else None
        if own and occ == occs[0] and impl == impls[0]:
          ref = chk
        print(os.environ.get('EARL_WS_TAG', 'ship'), f'rot={ROT} n={n} E={Ex} wgs/cu={occ} own={int(own)} impl={impl:3d}: {us:8.1f} us/launch  {us / Ex:6.2f} us/episode  {Ex * T * n * 66 / us / 1e6:6.2f} TB/s'
              + ('' if chk is None else f'  checksum {"ok" if chk == ref else "DIFF"}'), flush=True)
      finally:
        lib.earl_debug_set_rollout_impl(0)
lib.earl_debug_set_rollout_wgs_per_cu(1)
if os.environ.get('STAMPS'):
  lib.earl_debug_set_rollout_impl(29)
  for own in (True, False):
    a = acts[:E] if own else acts[0]
    o = tuple(t[:E] for t in out)
    for _ in range(3):
      env.rollout_episodes(a, episodes=E, out=o)
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16, np.uint64)
    lib.earl_debug_read_ws_profile(buf.ctypes.data, buf.size)
    b = buf.reshape(64, 16).astype(np.float64)
    m = np.median(b, axis=0)
    names = ['C.first_barrier', 'C.lds_read', 'C.compute', 'C.barrier', 'C.total', 'L.process+issue', 'L.barrier', 'L.total', 'S0.store', 'S0.barrier', 'S0.total',
             'Slast.store', 'Slast.barrier', 'C1.barrier', 'Llast.process', 'Llast.barrier']
    steps = 200 * ((E + 3) // 4)
    print(f'stamps own={int(own)} (median over workgroups 0..63 = group 0; ticks of s_memtime = 10 ns; per step of the group):')
    print('   ' + '  '.join(f'{nm} {m[k] / steps:.1f}' for k, nm in enumerate(names)))
  lib.earl_debug_set_rollout_impl(0)
