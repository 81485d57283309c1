"""where do the one-wave and the two-wave minitaur kernels part?  One TIMESTEP (num_substeps = 1: cold start) from identical states, per env; the states of the envs that differ
(and of as many that do not) are saved for a look on the CPU side.   python tools/mt_duo_bisect.py [steps]   (GPU) -> gpurun_out/mt_duo_bisect.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import ctypes as C
from earl_benchmark_amd import _abi
DBG = os.environ.get('MT_LIB')          # 'dbg': tools/ubench/libearl_mt_dbg.so (python tools/bench_mt_variant.py --build dbg -DEARL_MT_DEBUG): the active-set path per timestep
if DBG:
  _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_mt_{DBG}.so')
  if DBG == 'dbg':
    _abi.SIGNATURES['earl_debug_read_mt_dbg'] = [C.c_void_p, C.c_void_p]
from earl_benchmark_amd.envs.minitaur import Minitaur
lib = _abi.load()


def read_dbg():
  di, dd = np.zeros((4096, 8, 32), np.int32), np.zeros((14, 4096, 8, 32), np.float64)
  lib.earl_debug_read_mt_dbg(di.ctypes.data, dd.ctypes.data)
  return di, dd
n = 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sub = int(os.environ.get('SUBSTEPS', '1'))
lib.earl_debug_set_minitaur_duo(0)
a = Minitaur(num_envs=n, seed=1234, scalar_api=False); b = Minitaur(num_envs=n, seed=1234, scalar_api=False)
a.reset(); b.reset()
a._cfg.num_substeps = sub; b._cfg.num_substeps = sub
g = torch.Generator(device='cuda').manual_seed(99)
keys = ('qpos', 'qvel', 'overheat', 'motor_enabled', 'observed_torque', 'steps_since_reset')
saved = {'bad_q': [], 'bad_v': [], 'bad_act': [], 'bad_mp': [], 'ok_q': [], 'ok_v': [], 'ok_act': [], 'ok_mp': [], 'bad_dv': []}
tot = 0
for t in range(steps):
  act = (torch.rand(1, n, 8, generator=g, device='cuda') * 2 - 1).float()
  for k in keys:
    getattr(b, k).copy_(getattr(a, k))
  q0, v0 = a.qpos.clone(), a.qvel.clone()
  lib.earl_debug_set_minitaur_duo(0); a.rollout(act)
  torch.cuda.synchronize()
  d0 = read_dbg() if DBG == 'dbg' else None
  lib.earl_debug_set_minitaur_duo(1); b.rollout(act)
  torch.cuda.synchronize()
  d1 = read_dbg() if DBG == 'dbg' else None
  dq = (a.qvel - b.qvel).abs()
  bad = (dq.max(1).values > 0).nonzero().flatten()
  tot += len(bad)
  if len(bad):
    ok = torch.randperm(n, device='cuda')[:len(bad)]
    for name, idx in (('bad', bad), ('ok', ok)):
      saved[name + '_q'].append(q0[idx].cpu().numpy()); saved[name + '_v'].append(v0[idx].cpu().numpy()); saved[name + '_act'].append(act[0, idx].cpu().numpy())
      saved[name + '_mp'].append(a.motor_param[idx].cpu().numpy())
    saved['bad_dv'].append(dq[bad].cpu().numpy())
    print(f'step {t}: {len(bad)} envs differ: {bad.tolist()[:8]}  max |dqvel| {float(dq.max()):.2e}', flush=True)
    if DBG == 'dbg' and os.environ.get('FRAMES') and tot <= 12:
      for e in bad.tolist()[:3]:
        for ts in range(sub):
          A, B = d0[1][6:14, e, ts, :22], d1[1][6:14, e, ts, :22]
          if (A != B).any():
            for ee in (e, e ^ 1):
              print(f'   env {ee}: per timestep (contacts, warm, passes) ' + ' '.join(str(tuple(int(x) for x in d0[0][ee, k, :3])) for k in range(sub)), flush=True)
            for l in np.nonzero((A != B).any(0))[0]:
              print(f'   env {e} timestep {ts} dof {l}: q {A[7, l]!r}; Q mono {A[:4, l].tolist()} duo {B[:4, l].tolist()}; P mono {A[4:7, l].tolist()} duo {B[4:7, l].tolist()}', flush=True)
            break
    elif DBG == 'dbg' and tot <= 12:
      for e in bad.tolist()[:3]:
        for ts in range(sub):
          i0, i1 = d0[0][e, ts], d1[0][e, ts]
          dal = np.abs(d0[1][0, e, ts, :22] - d1[1][0, e, ts, :22]).max()
          dx = [float(np.abs(d0[1][1 + k, e, ts, :22] - d1[1][1 + k, e, ts, :22]).max()) for k in range(5)]
          ph = [float(np.abs(d0[1][6 + k, e, ts, :22] - d1[1][6 + k, e, ts, :22]).max()) for k in range(8)]
          phl = [int(np.abs(d0[1][6 + k, e, ts, :22] - d1[1][6 + k, e, ts, :22]).argmax()) for k in range(5)]
          print(f'   env {e} timestep {ts}: contacts {i0[0]}/{i1[0]} warm {i0[1]}/{i1[1]} passes {i0[2]}/{i1[2]} edges before {i0[4:4 + max(1, i0[0])].tolist()} / {i1[4:4 + max(1, i1[0])].tolist()} after '
                f'{i0[16:16 + max(1, i0[0])].tolist()} / {i1[16:16 + max(1, i1[0])].tolist()}  max |d qacc| {dal:.2e}; |d| of rw {dx[0]:.1e} Bw {dx[1]:.1e} Aw {dx[2]:.1e} qv_new {dx[3]:.1e} ext {dx[4]:.1e}; phases frames {ph[0]:.1e}@{phl[0]} K3 {ph[1]:.1e}@{phl[1]} K4 {ph[2]:.1e}@{phl[2]} K5 {ph[3]:.1e}@{phl[3]} tau {ph[4]:.1e}@{phl[4]} | inputs qp {ph[5]:.1e} root {ph[6]:.1e} qv {ph[7]:.1e}', flush=True)
print(f'total {tot} differing env-steps of {steps * n} ({sub} timestep(s) per step)')
lib.earl_debug_set_minitaur_duo(0)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'mt_duo_bisect.npz'), **{k: (np.concatenate(v) if v else np.zeros((0,))) for k, v in saved.items()})
