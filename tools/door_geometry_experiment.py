"""Round-4 experiment: alternative collision sets of the door model -- the handle's fat rod and base disc as BOXES with flat ends for the
finger corner points -- built in memory from the reference's MJCF facts (tools/mjcf_compile.py) and replayed against all ten recorded door episodes through the C
restatement (constants frozen).  Result: no effect (reverse 9.7 / 11.3 / 38.7 / 69.6 / 8.7 mm either way).  Run with `python -O` (the shipped loader asserts the
nv = 10 kernels' 16-block limit).  CPU, test infrastructure."""
import sys, os, json, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import mjcf_compile as mc
import heldout_eval as he
from oracle import physics_oracle as po, physics_c
physics_c.set_threads(8)
m = mc.compile_model('sawyer_door')
bw, dw = po.inverse_weights(po.Model(m)); m['body_invweight0'], m['dof_invweight0'] = bw, dw
pm = po.Model(m)
bp = pm.body_pos.copy(); bp[pm.body_id('door')] = np.array([0.1, 0.95, 0.1], np.float32).astype(float)
dl, frame, table = pm.body_id('door_link'), pm.body_id('doorlockB'), pm.body_id('tablelink')
colliding = lambda g: bool(m['geom_contype'][g] or m['geom_conaffinity'][g])
chains = [g for g in range(len(m['geom_body'])) if m['geom_body'][g] == dl and m['geom_type'][g] == 3 and colliding(g) and m['geom_size'][g][1] > m['geom_size'][g][0]]
big = [g for g in range(len(m['geom_body'])) if m['geom_body'][g] in (dl, frame, table) and m['geom_type'][g] == 4 and colliding(g)]
panel = [g for g in big if m['geom_body'][g] == dl][0]; tbl = [g for g in big if m['geom_body'][g] == table][0]
print('chains', [(g, m['geom_pos'][g].tolist(), m['geom_size'][g].tolist()) for g in chains])
P = [['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']]
def spec(variant):
  s = dict(plates=P, chains=[dict(geom=g, spacing=0.9 if m['geom_size'][g][0] < 0.015 else 0.75) for g in chains], corner_sets=P, big_boxes=big,
           edge_caps=dict(plates=P, caps=chains, per_plate=True), set_priority=('edge0', 'edge1'), set_cap=dict(edge0=1, edge1=1), drop_contained=True,
           drag=[(panel, tbl, -(bp[pm.body_id('door')][2] - m['geom_size'][panel][2]))], drag_calibration=mc.DOOR_DRAG_CALIBRATION)
  if 'sleeve' in variant:      # the fat rod (r 23 mm, half length 54 mm along x at (0.395, -0.12, 0) of door_link) as a box with FLAT ends for the finger corners
    s['explicit_boxes'] = [dict(body='door_link', pos=[0.395, -0.12, 0.0], half=[0.054, 0.0163, 0.0163], like=chains[-1], accept=('corner',), late=True)]
  if 'base' in variant:        # the handle's base disc (r 28 mm, half length 12 mm along y at (0.325, -0.006, 0)) as a box
    s.setdefault('explicit_boxes', []).append(dict(body='door_link', pos=[0.325, -0.006, 0.0], half=[0.0198, 0.012, 0.0198], like=chains[-1], accept=('corner',), late=True))
  if 'nodrop' in variant:
    s['drop_contained'] = False
  return s
class Rep(he.Replayer):
  def __init__(self, name):
    self.task = 'sawyer_door'
    self.cm = physics_c.CModel(name)
    s = self.cm.struct
    self.base_t, self.base_r = s.weld_invweight[0] / he.CAL_T, s.weld_invweight[1] / he.CAL_R
    self.base_G = np.array(s.drag_G[:]) / he.CAL_D
    self.hand = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
    self.cfg = physics_c.door_cfg(att_names=self.cm.att_names)
    self.eps = {d: he.episodes('sawyer_door', d) for d in ('forward', 'reverse')}
out = {}
for variant in sys.argv[1:] or ['ship', 'sleeve', 'sleeve+base']:
  red = po.reduce_model(pm, bp, attach_bodies=['hand'], attach_sites=['rightEndEffector', 'leftEndEffector', 'endEffector'], attach_geoms=['handle'], collision=spec(variant))
  path = os.path.join(ROOT, 'earl_benchmark_amd', 'models', 'sawyer_door_exp_links.npz')
  np.savez_compressed(path, **red)
  try:
    r = Rep('sawyer_door_exp'); r.set(he.CAL_T, he.CAL_R, he.CAL_D, 1)
    rows = r.replay('forward', list(range(5))) + r.replay('reverse', list(range(5)))
  finally:
    os.remove(path)
  sm = he.summarise(rows)
  print(variant, 'blocks', len(red['col_blk_begin']), 'pairs', len(red['col_pair']), json.dumps({k: (v['obj_rms_mm'], v['success']) for k, v in sm.items()}))
