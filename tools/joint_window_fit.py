#!/usr/bin/env python3
"""Round 6, the ONE bounded attempt VERDICT r05 item 4 asks for: re-identify the weld factors (and the start state) on PER-EPISODE contact-free windows -- every recorded
episode up to its own first touch (the step at which the recorded object first moves), not the fixed 13 / 38 / 11-step prefixes of round 4 -- fit set = even episodes,
held-out = odd ones; then the door-drag factor on the fit set's whole episodes, and the held-out protocol on whole episodes.

  python tools/joint_window_fit.py            -> profiles/r06_heldout_eval.json
Acceptance (VERDICT): hand error at first touch <= 0.5 mm on the held-out episodes AND more than 19 of 40 episodes reaching their recorded success with no group worse.
CPU only, test infrastructure (imports oracle/)."""
import ctypes as C
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import heldout_eval as he                                  # noqa: E402
import weld_free_motion_fit as wf                          # noqa: E402
from oracle import physics_c                               # noqa: E402
from oracle.tabletop_oracle import lib                     # noqa: E402

MARGIN = 2          # env steps kept clear of the first touch
TOUCH = 3e-4        # the recorded object has moved this far from where the episode found it: touched


def first_touch(ep):
  obj = ep[2][:, 4:7]
  moved = np.linalg.norm(obj - ep[0][4:7], axis=1) > TOUCH
  return int(np.argmax(moved)) if moved.any() else len(obj)


class Windows(wf.FreeMotion):
  """FreeMotion on per-episode windows: episode i is compared over its first n_i = first_touch - MARGIN env steps"""

  def __init__(self, task, which):
    super().__init__(task, which)
    self.sets = []
    self.meta = []
    for d in ('forward', 'reverse'):
      eps = he.episodes(task, d)
      sel = [(i, e) for i, e in enumerate(eps) if which == 'all' or i % 2 == (0 if which == 'fit' else 1)]
      n = [max(4, min(first_touch(e) - MARGIN, len(e[1]))) for _, e in sel]
      T = max(n)
      A = np.zeros((len(sel), T, 4)); R = np.zeros((len(sel), T, 3))
      for k, (_, e) in enumerate(sel):
        A[k, :n[k]] = np.clip(e[1][:n[k]].astype(np.float64), -1, 1); R[k, :n[k]] = e[2][:n[k], :3]
      self.sets.append((A, R, T, np.array(n)))
      self.meta += [(d, i, nk) for (i, _), nk in zip(sel, n)]

  def residuals(self, x, last_only=False):
    if len(x) > 14:
      self.set(np.exp(x[14]), np.exp(x[15]))
    cm = self.cm
    q1, v1 = self.qc.copy(), self.vc.copy(); q1[:7] = x[:7]; v1[:7] = x[7:14]
    st = cm.run(q1[None], v1[None], self.h0, [1, 0, 1, 0], [-1, 1], integrate=False)['att'][0, self.k]
    out = [] if last_only else [(st - self.o0) * 1e3 * 3]
    for A, R, T, n_ in self.sets:
      n = len(A); q, v = np.tile(q1, (n, 1)), np.tile(v1, (n, 1)); mp = np.tile(self.h0, (n, 1))
      for t in range(T):
        mp = np.clip(mp + A[:, t, :3] / 100, wf.LO, wf.HI)
        ct = np.stack([A[:, t, 3], -A[:, t, 3]], 1)
        r = cm.run(q, v, mp, [1, 0, 1, 0], ct, nsub=5); q, v = r['qpos'], r['qvel']
        hp = cm.run(q, v, mp, [1, 0, 1, 0], ct, integrate=False)['att'][:, self.k]
        live = (t == n_ - 1) if last_only else (t < n_)
        if last_only:
          for k in np.nonzero(live)[0]:
            out.append((hp[k] - R[k, t]) * 1e3)
        else:
          out.append((((hp - R[:, t]) * 1e3) * live[:, None]).ravel())
    return np.concatenate(out) if not last_only else np.stack(out)


def main():
  physics_c.set_threads(min(8, os.cpu_count() or 1))
  res = dict(protocol='per-episode contact-free windows (up to %d env steps before the recorded object first moves by %.1f mm); even episodes fit, odd held out' % (MARGIN, TOUCH * 1e3), tasks={})
  fits = {}
  # one (translation, rotation) pair for both tasks, a start state per task (as tools/weld_free_motion_fit.py --joint)
  fs = [Windows('sawyer_door', 'fit'), Windows('sawyer_peg', 'fit')]

  def resid(x):
    out = []
    for i, f in enumerate(fs):
      f.set(np.exp(x[28]), np.exp(x[29]))
      out += [f.residuals(x[14 * i:14 * i + 14]), wf.VEL_PENALTY * x[14 * i + 7:14 * i + 14]]
    return np.concatenate(out)
  x0 = np.concatenate([np.concatenate([f.cm.tables['reset_qpos_recorded'], f.cm.tables['reset_qvel_recorded']]) for f in fs] + [np.log([he.CAL_T, he.CAL_R])])
  e0 = resid(x0)
  sol = least_squares(resid, x0, diff_step=1e-4, x_scale=np.concatenate([np.full(7, 0.1), np.full(7, 1.0)] * 2 + [np.full(2, 0.3)]), max_nfev=150)
  ft, fr = float(np.exp(sol.x[28])), float(np.exp(sol.x[29]))
  print('window fit: translation x %.3f rotation x %.4f (shipped %.2f / %.3f)' % (ft, fr, he.CAL_T, he.CAL_R), flush=True)
  res['weld_translation'], res['weld_rotation'] = ft, fr
  for i, f in enumerate(fs):
    row = dict(windows={f'{d}/{k}': n for d, k, n in f.meta})
    for name, x, (t_, r_) in (('shipped', x0[14 * i:14 * i + 14], (he.CAL_T, he.CAL_R)), ('window_fit', sol.x[14 * i:14 * i + 14], (ft, fr))):
      for which in ('fit', 'heldout'):
        h = Windows(f.task, which); h.set(t_, r_)
        e = h.residuals(x)[3:]
        last = h.residuals(x, last_only=True)
        nlive = sum(n for _, _, n in h.meta) * 3
        row[f'{name}/{which}'] = dict(window_rms_mm=float(np.sqrt((e ** 2).sum() / nlive * 3)), hand_error_at_first_touch_mm=[[round(float(c), 2) for c in v] for v in last],
                                      hand_error_at_first_touch_norm_mm=[round(float(np.linalg.norm(v)), 2) for v in last])
        print(f.task, name, which, 'window rms %.2f mm; |error| at first touch' % row[f'{name}/{which}']['window_rms_mm'], row[f'{name}/{which}']['hand_error_at_first_touch_norm_mm'], flush=True)
    row['start_qpos'], row['start_qvel'] = sol.x[14 * i:14 * i + 7].tolist(), sol.x[14 * i + 7:14 * i + 14].tolist()
    res['tasks'][f.task] = row
    fits[f.task] = (sol.x[14 * i:14 * i + 7].copy(), sol.x[14 * i + 7:14 * i + 14].copy())
  # whole episodes: door drag chosen on the fit set, then fit / held-out / all with the shipped constants and with the window fit
  reps = [he.Replayer('sawyer_door'), he.Replayer('sawyer_peg')]
  shipped_state = {r.task: (np.array(r.cm.tables['reset_qpos_recorded']).copy(), np.array(r.cm.tables['reset_qvel_recorded']).copy()) for r in reps}

  def whole(params, state):
    for r in reps:
      r.cm.tables['reset_qpos_recorded'], r.cm.tables['reset_qvel_recorded'] = state[r.task]
    out = {}
    for which in ('fit', 'heldout'):
      rows = he.evaluate(reps, params, which)
      out[which] = dict(score_mm=round(1e3 * he.score(rows), 2), groups=he.summarise(rows))
    out['reached_of_40'] = int(sum(g['success'] for w in ('fit', 'heldout') for g in out[w]['groups'].values()))
    out['reached_by_group'] = {k: out['fit']['groups'][k]['success'] + out['heldout']['groups'][k]['success'] for k in out['fit']['groups']}
    return out
  res['whole_episodes'] = {}
  res['whole_episodes']['shipped (%.2f, %.3f, %.2f)' % (he.CAL_T, he.CAL_R, he.CAL_D)] = whole((he.CAL_T, he.CAL_R, he.CAL_D, 1), shipped_state)
  print('shipped:', res['whole_episodes'][list(res['whole_episodes'])[-1]]['reached_by_group'], flush=True)
  best = None
  for fd in (0.85, 0.9, 0.95, 1.0, 1.05):
    w = whole((ft, fr, fd, 1), fits)
    res['whole_episodes']['window fit (%.3f, %.4f), drag %.2f' % (ft, fr, fd)] = w
    print('window fit, drag', fd, 'fit score', w['fit']['score_mm'], 'held-out', w['heldout']['score_mm'], w['reached_by_group'], flush=True)
    if best is None or w['fit']['score_mm'] < best[1]:
      best = (fd, w['fit']['score_mm'])
  res['door_drag_chosen_on_fit_set'] = best[0]
  for r in reps:
    r.cm.tables['reset_qpos_recorded'], r.cm.tables['reset_qvel_recorded'] = shipped_state[r.task]
    r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
  json.dump(res, open(os.path.join(ROOT, 'profiles', 'r06_heldout_eval.json'), 'w'), indent=1)


if __name__ == '__main__':
  main()
