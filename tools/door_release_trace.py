#!/usr/bin/env python3
"""Round-4 contact-level trace of a recorded door episode: the C restatement replays the episode open loop up to env step T0, the numpy statement then
steps on to T1 printing per env step the replayed and recorded hand, the mocap, the finger joints, the hand's tilt against the mocap orientation, the six weld forces and
every contact (pair, links, depth, normal, point).  Used to see why reverse episodes 2 and 3 keep the rod after the gripper opens (the +y finger sits at the frictionless
equilibrium height behind the rod; in the recording it has climbed on top of it).      python tools/door_release_trace.py [episode=3] [T0=76] [T1=88] [forward|reverse]
CPU, test infrastructure (imports oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import heldout_eval as he                                  # noqa: E402
from oracle import physics_c, physics_oracle as po         # noqa: E402

np.set_printoptions(precision=1, suppress=True, linewidth=250)
args = [a for a in sys.argv[1:] if a.lstrip('-').isdigit()]
ep, T0, T1 = (int(a) for a in (args + ['3', '76', '88'][len(args):])[:3])
direction = 'forward' if 'forward' in sys.argv else 'reverse'
physics_c.set_threads(min(8, os.cpu_count() or 1))
r = he.Replayer('sawyer_door'); r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
cm = r.cm; e = r.eps[direction][ep]
q0, v0 = r.settled(); q, v, mp = q0[None].copy(), v0[None].copy(), r.hand[None].copy()
k = cm.att_names.index('handle')
angs = np.linspace(-1.5, 0.1, 1601); qq = np.tile(q0, (len(angs), 1)); qq[:, 9] = angs
att = cm.run(qq, np.zeros_like(qq), r.hand, [1, 0, 1, 0], [-1, 1], integrate=False)['att'][:, k]
q[0, 9] = angs[int(np.argmin(((att - e[0][4:7]) ** 2).sum(1)))]; v[0, 9] = 0
obs, _, _, _ = cm.sawyer_rollout(r.cfg, q, v, mp, e[0][7:][None], np.zeros(1, np.int32), e[1][:T0, None, :].astype(np.float32))
print(f'{direction} episode {ep}: state after {T0} env steps: hand', obs[-1, 0, :3] * 1e3, 'recorded', e[2][T0 - 1, :3] * 1e3, 'mocap', mp[0] * 1e3, 'door angle %.4f' % q[0, 9])
lm = po.LinkModel(os.path.join(ROOT, 'earl_benchmark_amd', 'models', 'sawyer_door_links.npz'))
qn, vn, mpn = q[0].copy(), v[0].copy(), mp[0].copy()
lo, hi = np.array([-0.5, 0.40, 0.05]), np.array([0.5, 1.0, 0.5])
hk = cm.att_names.index('hand')
for t in range(T0, min(T1, len(e[1]))):
  a = np.clip(e[1][t].astype(float), -1, 1)
  mpn = np.clip(mpn + a[:3] / 100, lo, hi)
  ap = None
  for _ in range(5):
    qn, vn, out = lm.step(qn, vn, [a[3], -a[3]], mpn, [1.0, 0, 1.0, 0], a_prev=ap); ap = out['qacc']
  hp, hq = lm.attachment(out['pos'], out['quat'], hk)
  tilt = 2 * np.degrees(np.arcsin(min(1, np.linalg.norm(po.quat_mul(po.quat_conj(hq), np.array([1, 0, 1, 0]) / np.sqrt(2))[1:]))))
  print(t, 'hand', hp * 1e3, 'recorded', e[2][t, :3] * 1e3, 'mocap', mpn * 1e3, 'fingers', qn[7:9] * 1e3, 'tilt %.1f deg' % tilt, 'weld f', out['f'][:6])
  for c in out['contacts']:
    print('     contact pair', c['pair'], 'cls', c['cls'], 'links', c['ls'], c['lb'], 'depth %.2f mm' % (-c['dist'] * 1e3), 'n', np.round(c['n'], 2), 'p', c['p'] * 1e3)
