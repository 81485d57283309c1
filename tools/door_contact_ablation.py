#!/usr/bin/env python3
"""the door's gripper-handle contacts as the reference models them, ablated on the C restatement (no GPU): all ten recorded door episodes replayed
open loop (tools/heldout_eval.py's Replayer) with each collision table variant of tools/mjcf_compile.py (DOOR_CONTACTS=...): 'chains' = the shipped round-4 set,
'cyl' = the handle's four cylinders as cylinders with ONE contact per (box, cylinder) pair (portal refinement, oracle/physics_oracle.c mpr_box_cylinder),
'+split' = claw plate and pad as separate boxes with their own parameters, '+tor' = condim-4 torsional rows.  Constants 3.35 / 0.07 / 0.95 frozen.
  python tools/door_contact_ablation.py [variant ...]  -> table; --json PATH writes it"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import heldout_eval as he                        # noqa: E402
from oracle import physics_c                    # noqa: E402


def run(variant, offset=(0.0, 0.0, 0.0)):
  """offset: the whole mocap path shifted by this much (the reset pose is not): how much of the outcome is decided by millimetres of the approach"""
  name = 'sawyer_door' if variant == 'chains' else 'sawyer_door_' + variant.replace('+', '_')
  r = he.Replayer.__new__(he.Replayer)
  r.task = 'sawyer_door'
  r.cm = physics_c.CModel(name)
  s = r.cm.struct
  r.base_t, r.base_r = s.weld_invweight[0] / he.CAL_T, s.weld_invweight[1] / he.CAL_R
  r.base_G = np.array(s.drag_G[:]) / he.CAL_D
  r.hand = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
  r.cfg = physics_c.door_cfg(att_names=r.cm.att_names)
  r.eps = {d: he.episodes('sawyer_door', d) for d in ('forward', 'reverse')}
  r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
  if any(offset):
    q0v0 = r.settled()
    r.settled = lambda: (q0v0[0].copy(), q0v0[1].copy())
    r.hand = r.hand + np.asarray(offset, float)
  rows = {d: r.replay(d, list(range(5))) for d in ('forward', 'reverse')}
  return {d: dict(obj_rms_mm=[round(1e3 * x['obj_rms'], 1) for x in v], final_mm=[round(1e3 * x['obj_final'], 1) for x in v], hand_rms_mm=[round(1e3 * x['hand_rms'], 1) for x in v],
                  reached=int(sum(x['success'] for x in v)), recorded=int(sum(x['recorded_success'] for x in v))) for d, v in rows.items()}


def main():
  physics_c.set_threads(min(8, os.cpu_count() or 1))
  args = [a for i, a in enumerate(sys.argv[1:]) if not a.startswith('--') and sys.argv[i] != '--json']
  out = {}
  for v in args or ['chains', 'cyl', 'cyl+split', 'cyl+split+tor']:
    out[v] = run(v)
    f, b = out[v]['forward'], out[v]['reverse']
    print(f'{v:16s} forward {f["obj_rms_mm"]} reached {f["reached"]}/5 | reverse {b["obj_rms_mm"]} reached {b["reached"]}/5', flush=True)
  if '--offsets' in sys.argv:                   # sensitivity: the mocap path shifted by (dy, dz) millimetres
    out['mocap_path_offsets_mm'] = {}
    for v in ('chains', 'cyl'):
      for dy, dz in ((0, -2), (0, -4), (2, -2), (4, -4)):
        res = run(v, (0.0, dy * 1e-3, dz * 1e-3))
        out['mocap_path_offsets_mm'][f'{v} dy {dy} dz {dz}'] = res
        print(f'{v:8s} dy {dy} dz {dz}: forward {res["forward"]["obj_rms_mm"]} {res["forward"]["reached"]}/5 | reverse {res["reverse"]["obj_rms_mm"]} {res["reverse"]["reached"]}/5', flush=True)
  if '--json' in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)


if __name__ == '__main__':
  main()
