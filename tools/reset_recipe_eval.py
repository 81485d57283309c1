#!/usr/bin/env python3
"""should the reference's literal reset recipe (sim.reset() + _reset_hand = 250 timesteps, sawyer_door.py:111-125) be the DEFAULT start state of
the Sawyer envs instead of the converged pose (2000 timesteps)?  All 40 recorded episodes replayed open loop from both start states through the C restatement
(constants frozen: weld 4.0 / rule / drag 0.8), object-path RMS per episode -> profiles/r04_reset_recipe_eval.json.  CPU, test infrastructure."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import heldout_eval as he
from oracle import physics_c
physics_c.set_threads(8)
res = {}
for nsub in (250, 2000):
  rows = []
  for task in ('sawyer_door', 'sawyer_peg'):
    r = he.Replayer(task)
    r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
    def settled(self=r, nsub=nsub):
      cm = self.cm
      q0 = cm.tables['qpos0'][None] if self.task == 'sawyer_peg' else np.zeros((1, cm.nv))
      rr = cm.run(q0, np.zeros((1, cm.nv)), self.hand, [1, 0, 1, 0], [-1, 1], nsub=nsub)
      return rr['qpos'][0].copy(), rr['qvel'][0].copy()
    r.settled = settled
    for d in ('forward', 'reverse'):
      rows += r.replay(d, list(range(len(r.eps[d]))))
  res[nsub] = he.summarise(rows)
  print(nsub, json.dumps(res[nsub]))
json.dump({'start_state_timesteps': {str(k): v for k, v in res.items()}, 'constants': 'weld translation 4.0, rotation rule (raw mocap quaternion), door drag 0.8 (frozen)',
           'metric': 'RMS distance between replayed and recorded object path (door handle / pegHead), open loop over the whole episode, mm'},
          open(os.path.join(ROOT, 'profiles', 'r04_reset_recipe_eval.json'), 'w'), indent=1)
