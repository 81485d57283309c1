"""Round-4 observation: the door arm's state along sim.reset() + N timesteps of _reset_hand, with the shipped (x 4) and the derived (x 1) translational
weld weight: hand position, joint angles, residual speed, weld forces, limit forces.  At the reference's N = 250 the arm is mid-transient (0.5 rad/s, weld force 45 - 52 N,
no joint on a limit); converged (N >= 2000) right_j1 rests on its upper limit and the weld carries 31 N.  CPU, test infrastructure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from oracle import physics_c
from oracle.tabletop_oracle import lib
np.set_printoptions(precision=4, suppress=True, linewidth=200)
cm = physics_c.CModel('sawyer_door')
s = cm.struct
t = cm.tables
print('link masses', np.round(t['mass'], 3), 'sum arm', t['mass'][:9].sum())
print('ranges j0..j8', t['jnt_range'][:9].tolist())
print('weld_invweight (shipped)', s.weld_invweight[0], s.weld_invweight[1])
hand = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
lib().oracle_set_raw_mocap_quat(C.c_int(1))
def settle(nsub, iw_scale=1.0):
  base = s.weld_invweight[0]
  s.weld_invweight[0] = base * iw_scale
  r = cm.run(np.zeros((1, cm.nv)), np.zeros((1, cm.nv)), hand, [1, 0, 1, 0], [-1, 1], nsub=nsub)
  f = cm.run(r['qpos'], r['qvel'], hand, [1, 0, 1, 0], [-1, 1], integrate=False)
  s.weld_invweight[0] = base
  k = cm.att_names.index('hand')
  return r['qpos'][0], r['qvel'][0], f['att'][0, k], f['efc'][0]
for nsub in (250, 500, 1000, 2000, 8000):
  for sc in (1.0, 0.25):
    q, v, hp, efc = settle(nsub, sc)
    print(f'nsub {nsub:5d} weld iw x{sc}: hand {np.round(hp*1e3,2)} mm  q {np.round(q[:7],3)} |qvel| {np.abs(v[:7]).max():.3f}  weld f {np.round(efc[:6],2)}  limit forces {np.round(efc[6:6+14],1)}')
