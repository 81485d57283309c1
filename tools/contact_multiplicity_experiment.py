#!/usr/bin/env python3
"""Round-4 experiment (not adopted): divide the regulariser of the gripper-plate contact classes by n ("n contact points where this build has one": MuJoCo's
box-box routine returns up to eight per finger, and with mu = 2 the four pyramid edges make the normal direction 2 x softer than the elliptic cone) and replay all recorded
episodes of the task.  Measured: peg n = 2: path RMS 4.7 / 5.4 mm, goals 4 / 10 forward, 13 / 20 reverse (n = 1: 4.9 / 5.5, 2, 8; n = 4: 6.1 / 6.5, 6, 17); door n = 2: reverse
episode 4 breaks (9.0 -> 57 mm).      python tools/contact_multiplicity_experiment.py sawyer_peg 1,2,4
CPU, test infrastructure (imports oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import heldout_eval as he                                  # noqa: E402
from oracle import physics_c                               # noqa: E402

physics_c.set_threads(min(8, os.cpu_count() or 1))
task = sys.argv[1] if len(sys.argv) > 1 else 'sawyer_peg'
r = he.Replayer(task); r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
classes = [c for c in range(r.cm.col.n_cls) if r.cm.col.cls_mu[c] == 2.0]          # the pad classes (friction 2): plates against peg / handle
base = np.array(r.cm.col.cls_invw[:r.cm.col.n_cls])
for n in [float(a) for a in (sys.argv[2] if len(sys.argv) > 2 else '1,2,4').split(',')]:
  for c in classes:
    r.cm.col.cls_invw[c] = base[c] / n
  out = []
  for d in ('forward', 'reverse'):
    rows = r.replay(d, list(range(len(r.eps[d]))))
    out.append(f"{d}: object-path RMS mean {np.mean([1e3 * x['obj_rms'] for x in rows]):.1f} mm {[round(1e3 * x['obj_rms'], 1) for x in rows]} goals {sum(x['success'] for x in rows)} / {len(rows)}")
  print(f'{task} multiplicity {n:g} on classes {classes} |', ' | '.join(out), flush=True)
