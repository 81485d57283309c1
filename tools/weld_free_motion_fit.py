#!/usr/bin/env python3
"""Round-4 identification of the mocap weld on the CONTACT-FREE prefixes of the recorded Sawyer episodes.

Before the gripper touches anything, the recorded hand path depends only on the arm (MJCF masses, joint damping 10, armature), the weld rows and the
state the episode started from (the reference's sim.reset() + 250 timesteps, which no stepper can reproduce: violent transient from qpos0 with right_j1 outside
its range).  So the prefixes identify the weld by themselves.  Unknowns of one fit: the arm's start state (7 angles, 7 speeds; the same for every episode of a task,
as in the reference) and log factors on the translational / rotational regulariser; residuals: recorded hand position - replayed one over the first N env steps of
the FIT-set episodes (even-numbered, tools/heldout_eval.py's split) plus the recorded first observation.  Replays run on the C restatement without contacts.

  python tools/weld_free_motion_fit.py sawyer_door            # joint fit -> profiles/r04_weld_free_motion_fit.json (door and peg separately; they agree)
  python tools/weld_free_motion_fit.py sawyer_door --scan     # RMS over a grid of factors, start state re-fitted at every point
  python tools/weld_free_motion_fit.py sawyer_door --axes     # six per-row factors free (oracle_set_weld_row_scale): is the translational factor isotropic?
  python tools/weld_free_motion_fit.py sawyer_door --damping  # the two weld factors AND the arm's joint damping (joints 0-3, 4-6) free: is the arm model what absorbs the factors?
  python tools/weld_free_motion_fit.py --joint                # one pair for both tasks: the shipped values

CPU, test infrastructure (imports oracle/); reads the demonstrations shipped under earl_benchmark_amd/demonstrations."""
import ctypes as C
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import heldout_eval as he                                  # noqa: E402
from oracle import physics_c                               # noqa: E402
from oracle.tabletop_oracle import lib                     # noqa: E402

LO, HI = np.array([-0.5, 0.40, 0.05]), np.array([0.5, 1.0, 0.5])
PREFIX = {('sawyer_door', 'forward'): 13, ('sawyer_door', 'reverse'): 38, ('sawyer_peg', 'forward'): 11, ('sawyer_peg', 'reverse'): 11}   # env steps before the first contact


class FreeMotion:
  def __init__(self, task, which='fit'):
    self.task = task
    self.cm = physics_c.CModel(task, contacts=False)
    s = self.cm.struct
    self.derived = (s.weld_invweight[0] / he.CAL_T, s.weld_invweight[1] / he.CAL_R)
    self.h0 = np.array([0, .4, .2] if task == 'sawyer_door' else [0, .6, .2], np.float32).astype(float)
    self.k = self.cm.att_names.index('hand')
    self.sets = []
    for d in ('forward', 'reverse'):
      eps = he.episodes(task, d); n = PREFIX[(task, d)]
      eps = [e for i, e in enumerate(eps) if which == 'all' or i % 2 == (0 if which == 'fit' else 1)]
      self.sets.append((np.stack([np.clip(e[1][:n].astype(np.float64), -1, 1) for e in eps]), np.stack([e[2][:n, :3] for e in eps]), n))
    self.o0 = he.episodes(task, 'forward')[0][0][:3]
    lib().oracle_set_raw_mocap_quat(C.c_int(1))
    cm = self.cm
    q00 = cm.tables['qpos0'][None] if task == 'sawyer_peg' else np.zeros((1, cm.nv))
    r = cm.run(q00, np.zeros((1, cm.nv)), self.h0, [1, 0, 1, 0], [-1, 1], nsub=2000)
    self.qc, self.vc = r['qpos'][0].copy(), r['qvel'][0].copy()

  def set(self, ft, fr):
    s = self.cm.struct
    s.weld_invweight[0], s.weld_invweight[1] = self.derived[0] * ft, self.derived[1] * fr

  def residuals(self, x):
    """x = 7 arm angles, 7 arm speeds [, log ft, log fr] -> mm"""
    if len(x) > 14:
      self.set(np.exp(x[14]), np.exp(x[15]))
    cm = self.cm
    q1, v1 = self.qc.copy(), self.vc.copy(); q1[:7] = x[:7]; v1[:7] = x[7:14]
    st = cm.run(q1[None], v1[None], self.h0, [1, 0, 1, 0], [-1, 1], integrate=False)['att'][0, self.k]
    out = [(st - self.o0) * 1e3 * 3]
    for A, R, n_steps in self.sets:
      n = len(A); q, v = np.tile(q1, (n, 1)), np.tile(v1, (n, 1)); mp = np.tile(self.h0, (n, 1))
      for t in range(n_steps):
        mp = np.clip(mp + A[:, t, :3] / 100, LO, HI)
        ct = np.stack([A[:, t, 3], -A[:, t, 3]], 1)
        r = cm.run(q, v, mp, [1, 0, 1, 0], ct, nsub=5); q, v = r['qpos'], r['qvel']
        hp = cm.run(q, v, mp, [1, 0, 1, 0], ct, integrate=False)['att'][:, self.k]
        out.append(((hp - R[:, t]) * 1e3).ravel())
    return np.concatenate(out)

  def fit(self, ft=None, fr=None, x0=None):
    free = ft is None
    x = np.concatenate([self.qc[:7], self.vc[:7]]) if x0 is None else np.asarray(x0, float)[:14]
    if free:
      x = np.concatenate([x, np.log([3.5, 0.1])])
    else:
      self.set(ft, fr)
    res = least_squares(self.residuals, x, diff_step=1e-4, x_scale=np.concatenate([np.full(7, 0.1), np.full(7, 1.0), np.full(len(x) - 14, 0.3)]), max_nfev=150)
    e = self.residuals(res.x)[3:]
    return res.x, float(np.sqrt((e ** 2).mean() * 3))


VEL_PENALTY = 0.2       # mm of residual per rad/s of start speed: keeps the speeds of the low-inertia wrist joints (which decay within a timestep and are not observable) near zero


def joint_fit():
  """one (translation, rotation) pair for both tasks, a start state per task; fit-set episodes only -> dict"""
  fs = [FreeMotion('sawyer_door', 'fit'), FreeMotion('sawyer_peg', 'fit')]
  def res(x):
    out = []
    for i, f in enumerate(fs):
      f.set(np.exp(x[28]), np.exp(x[29]))
      out += [f.residuals(x[14 * i:14 * i + 14]), VEL_PENALTY * x[14 * i + 7:14 * i + 14]]
    return np.concatenate(out)
  x0 = np.concatenate([np.concatenate([f.qc[:7], f.vc[:7]]) for f in fs] + [np.log([3.5, 0.1])])
  sol = least_squares(res, x0, diff_step=1e-4, x_scale=np.concatenate([np.full(7, 0.1), np.full(7, 1.0)] * 2 + [np.full(2, 0.3)]), max_nfev=200)
  x = sol.x; ft, fr = float(np.exp(x[28])), float(np.exp(x[29]))
  out = dict(weld_translation=ft, weld_rotation=fr, velocity_penalty_mm_per_rad_s=VEL_PENALTY, tasks={})
  for i, f in enumerate(fs):
    f.set(ft, fr)
    e = f.residuals(x[14 * i:14 * i + 14]); first, e = e[:3] / 3, e[3:]
    h = FreeMotion(f.task, 'heldout'); h.set(ft, fr)
    eh = h.residuals(x[14 * i:14 * i + 14])[3:]
    out['tasks'][f.task] = dict(fit_rms_mm=float(np.sqrt((e ** 2).mean() * 3)), heldout_rms_mm=float(np.sqrt((eh ** 2).mean() * 3)), first_observation_error_mm=first.tolist(),
                                start_qpos=x[14 * i:14 * i + 7].tolist(), start_qvel=x[14 * i + 7:14 * i + 14].tolist())
    print(f.task, out['tasks'][f.task], flush=True)
  print('joint fit: translation x %.3f, rotation x %.4f of the derived values' % (ft, fr))
  return out


def axes_fit(task):
  """the six weld rows with a factor each (on top of the derived values): translation x, y, z (world axes), rotation x, y, z (hand axes)"""
  f = FreeMotion(task, 'all'); f.set(1.0, 1.0)
  def res(x):
    lib().oracle_set_weld_row_scale((C.c_double * 6)(*np.exp(x[14:20])))
    return f.residuals(x[:14])
  x0 = np.concatenate([f.qc[:7], f.vc[:7], np.log([3.35] * 3 + [0.07] * 3)])
  try:
    sol = least_squares(res, x0, diff_step=1e-4, x_scale=np.concatenate([np.full(7, 0.1), np.full(7, 1.0), np.full(6, 0.3)]), max_nfev=200)
    e = res(sol.x)[3:]
  finally:
    lib().oracle_set_weld_row_scale(None)
  out = dict(row_factors=np.exp(sol.x[14:20]).tolist(), rms_mm=float(np.sqrt((e ** 2).mean() * 3)))
  print(task, 'per-row factors (translation x y z, rotation x y z)', np.round(out['row_factors'], 3), 'RMS %.2f mm (all episodes)' % out['rms_mm'])
  return out


def damping_fit(task):
  """the weld factors and two factors on the MJCF's joint damping (10 N m s / rad on every arm joint): proximal joints 0-3, wrist joints 4-6"""
  f = FreeMotion(task, 'all'); st = f.cm.struct
  base = np.array(st.damping[:7])
  def res(x):
    for j in range(7):
      st.damping[j] = base[j] * np.exp(x[16 if j < 4 else 17])
    return f.residuals(x[:16])
  x0 = np.concatenate([f.qc[:7], f.vc[:7], np.log([3.35, 0.07]), [0.0, 0.0]])
  try:
    sol = least_squares(res, x0, diff_step=1e-4, x_scale=np.concatenate([np.full(7, 0.1), np.full(7, 1.0), np.full(4, 0.3)]), max_nfev=200)
    e = res(sol.x)[3:]
  finally:
    for j in range(7):
      st.damping[j] = base[j]
  out = dict(weld_translation=float(np.exp(sol.x[14])), weld_rotation=float(np.exp(sol.x[15])), damping_proximal=float(np.exp(sol.x[16])), damping_wrist=float(np.exp(sol.x[17])),
             rms_mm=float(np.sqrt((e ** 2).mean() * 3)))
  print(task, out)
  return out


def forms():
  """candidate FORMS of the weld's rotational rows and regularisers, each at factor 1.0 (nothing fitted but the start state, which every candidate gets
  re-fitted for itself), on the contact-free prefixes: fit-set RMS and held-out RMS per task -> profiles/r05_weld_forms.json.  The row algebra lives in oracle/physics_oracle.py
  (constraints()) / oracle/physics_oracle.c; a form is selected here through the restatement's switches: the mocap quaternion as given or normalised, a scale on the rotational
  rows (rows x s == regulariser / s^2), the two regularisers."""
  cands = [
      ('shipped: rows = vec(conj(q_hand) q_mocap) with the mocap quaternion as given [1, 0, 1, 0], R x (3.35, 0.07) [calibrated]', dict(raw=1, ft=he.CAL_T, fr=he.CAL_R, rot_scale=1.0)),
      ('A  same rows, mocap quaternion as given, derived regularisers (factors 1, 1)', dict(raw=1, ft=1.0, fr=1.0, rot_scale=1.0)),
      ('B  MuJoCo as documented: mocap quaternion normalised (mj_kinematics), rows = vector part of the difference quaternion (~ theta / 2), Jacobian 0.5 x angular, '
       'R = (1 - d) / d x body_invweight0[hand] (translational / rotational)', dict(raw=0, ft=1.0, fr=1.0, rot_scale=1.0)),
      ('C  rotation VECTOR rows (theta x axis, angular Jacobian unscaled) = B with the rotational rows x 2', dict(raw=0, ft=1.0, fr=1.0, rot_scale=2.0)),       # rot_scale = factor on the ROWS (applied below as 1 / s^2 on the regulariser)
      ('D  B with the inverse weights from J diag(dof_invweight0) J^T at qpos0 (translation x 3.52, rotation x 1.2 of B)', dict(raw=0, ft=3.52, fr=1.2, rot_scale=1.0)),
      ('E  B with ONE regulariser for all six rows: the translational weight', dict(raw=0, ft=1.0, fr='tran', rot_scale=1.0)),
      ('F  B with ONE regulariser for all six rows: the rotational weight', dict(raw=0, ft='rot', fr=1.0, rot_scale=1.0)),
      ('G  rotational rows x 4 of B (regulariser / 16 = 0.0625: what the calibrated 0.07 amounts to as a row scale), translational rows as B', dict(raw=0, ft=1.0, fr=1.0, rot_scale=4.0)),
      ('H  D for the translational rows (x 3.52) and G for the rotational ones (rows x 4): two RULES, nothing calibrated', dict(raw=0, ft=3.52, fr=1.0, rot_scale=4.0)),
      ('I  D for the translational rows and C for the rotational ones (rotation vector: rows x 2)', dict(raw=0, ft=3.52, fr=1.0, rot_scale=2.0)),
      ('J  D for the translational rows, rows x 2 sqrt(2) (rotation vector with the mocap quaternion as given)', dict(raw=0, ft=3.52, fr=1.0, rot_scale=2.8284271247461903)),
  ]
  out = {}
  for name, c in cands:
    row = {}
    for task in ('sawyer_door', 'sawyer_peg'):
      f = FreeMotion(task, 'fit')
      lib().oracle_set_raw_mocap_quat(C.c_int(c['raw']))
      rs = 1.0 / c['rot_scale'] ** 2                        # oracle_set_weld_row_scale multiplies the REGULARISER of a row: rows x s == regulariser / s^2
      sc = (C.c_double * 6)(1, 1, 1, rs, rs, rs)
      lib().oracle_set_weld_row_scale(sc)
      dt_, dr_ = f.derived
      ft = c['ft'] if c['ft'] != 'rot' else dr_ / dt_
      fr = c['fr'] if c['fr'] != 'tran' else dt_ / dr_
      try:
        x, rms = f.fit(ft, fr)
        h = FreeMotion(task, 'heldout'); h.set(ft, fr)
        lib().oracle_set_raw_mocap_quat(C.c_int(c['raw'])); lib().oracle_set_weld_row_scale(sc)
        e = h.residuals(x[:14])[3:]
        row[task] = dict(fit_rms_mm=round(rms, 2), heldout_rms_mm=round(float(np.sqrt((e ** 2).mean() * 3)), 2))
      finally:
        lib().oracle_set_weld_row_scale(None); lib().oracle_set_raw_mocap_quat(C.c_int(1))
    out[name] = row
    print(f"{name[:110]:110s} door fit {row['sawyer_door']['fit_rms_mm']:5.2f} held-out {row['sawyer_door']['heldout_rms_mm']:5.2f} | peg fit {row['sawyer_peg']['fit_rms_mm']:5.2f} held-out "
          f"{row['sawyer_peg']['heldout_rms_mm']:5.2f} mm", flush=True)
  json.dump(out, open(os.path.join(ROOT, 'profiles', 'r05_weld_forms.json'), 'w'), indent=1)


def main():
  physics_c.set_threads(min(8, os.cpu_count() or 1))
  if '--forms' in sys.argv:
    return forms()
  if '--damping' in sys.argv:
    task = ([a for a in sys.argv[1:] if not a.startswith('-')] or ['sawyer_door'])[0]
    path = os.path.join(ROOT, 'profiles', 'r04_weld_free_motion_fit.json')
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev.setdefault('damping_free', {})[task] = damping_fit(task)
    json.dump(prev, open(path, 'w'), indent=1)
    return
  if '--axes' in sys.argv:
    task = ([a for a in sys.argv[1:] if not a.startswith('-')] or ['sawyer_door'])[0]
    path = os.path.join(ROOT, 'profiles', 'r04_weld_free_motion_fit.json')
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev.setdefault('per_row', {})[task] = axes_fit(task)
    json.dump(prev, open(path, 'w'), indent=1)
    return
  if '--joint' in sys.argv:
    path = os.path.join(ROOT, 'profiles', 'r04_weld_free_motion_fit.json')
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev['joint'] = joint_fit()
    json.dump(prev, open(path, 'w'), indent=1)
    return
  task = [a for a in sys.argv[1:] if not a.startswith('-')][0] if any(not a.startswith('-') for a in sys.argv[1:]) else 'sawyer_door'
  f = FreeMotion(task, 'fit')
  x_cal, rms_cal = f.fit(he.CAL_T, he.CAL_R)
  print(task, f'shipped (x {he.CAL_T}, rule): start state free, fit-set prefixes RMS {rms_cal:.2f} mm')
  x, rms = f.fit()
  ft, fr = float(np.exp(x[14])), float(np.exp(x[15]))
  print(task, f'fitted: translation x {ft:.3f}, rotation x {fr:.4f} of the derived values, RMS {rms:.2f} mm; start q {np.round(x[:7], 4).tolist()} v {np.round(x[7:14], 3).tolist()}')
  h = FreeMotion(task, 'heldout'); h.set(ft, fr)
  e = h.residuals(x[:14])[3:]
  print(task, f'held-out prefixes with these values and this start state: RMS {np.sqrt((e ** 2).mean() * 3):.2f} mm')
  out = dict(task=task, shipped_rms_mm=rms_cal, weld_translation=ft, weld_rotation=fr, fit_rms_mm=rms, heldout_rms_mm=float(np.sqrt((e ** 2).mean() * 3)),
             start_qpos=x[:7].tolist(), start_qvel=x[7:14].tolist(), prefix_steps={f'{k[0]}/{k[1]}': v for k, v in PREFIX.items() if k[0] == task})
  if '--scan' in sys.argv:
    out['scan'] = []
    for ft_ in (2.0, 3.0, 3.516, 4.0, 5.0):
      for fr_ in (0.03, 0.05, 0.07, 0.1, 0.15, 0.3, 1.0):
        _, r_ = f.fit(ft_, fr_, x)
        out['scan'].append(dict(weld_translation=ft_, weld_rotation=fr_, rms_mm=r_)); print(out['scan'][-1], flush=True)
  path = os.path.join(ROOT, 'profiles', 'r04_weld_free_motion_fit.json')
  prev = json.load(open(path)) if os.path.exists(path) else {}
  prev[task] = out
  json.dump(prev, open(path, 'w'), indent=1)


if __name__ == '__main__':
  main()
