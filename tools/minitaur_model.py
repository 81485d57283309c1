#!/usr/bin/env python3
"""Author the minitaur model tables (earl_benchmark_amd/models/minitaur_links.npz) for the articulated-body stepper.

The reference drives PyBullet with `pybullet_data/quadruped/minitaur.urdf`, which is NOT in the reference tree: there is no
model file to compile.  What the reference's own Python does state about the robot is used as FACT (cited below); everything else -- link dimensions,
masses, inertias, contact geometry -- is THIS BUILD'S OWN AUTHORING of a Ghost-Minitaur-like quadruped, declared as such in DESIGN.md section 14.
**Parity with the reference's Bullet simulation is unpinned and model-less.**

Facts taken from the reference (file:line under earl_benchmark/envs/):
  * structure: one floating base + 4 legs x (2 motor joints + 2 knee joints) = 16 revolute joints, link-id tables  minitaur.py:16-25
  * leg order front_left, back_left, front_right, back_right; motor order L, R per leg                              minitaur.py:17-24
  * motor direction -1 for the left legs' motors, +1 for the right legs'                                            minitaur.py:80
  * reset pose: base at (0, 0, 0.2), identity orientation; motor joints at dir * pi/2, knee joints at dir * -2.1834 minitaur.py:10-11, 187-211
  * loop closure: per leg a point-to-point constraint between the R knee link at (0, 0.005, 0.2) and the L knee
    link at (0, 0.01, 0.2) of their own frames -> the lower legs are 0.2 long along their local z                   minitaur.py:12-13, 212-217
  * gravity (0, 0, -10), time step 0.01 / 5 = 0.002, 5 simulation steps per env step                                minitaur_gym_env.py:25, 126-128, 161-164, 232
  * ground plane + 12 wall tiles: boxes 0.10 x 1.00 x 0.5 centred 0.05 behind the tile origin at height 0.5 (so they
    span z in [0.25, 0.75]), three per side at +-1.5                                                                minitaur_gym_env.py:39-50, 210-220, minitaur_assets/wall_tile.urdf:19-24
Own authoring: base 3.3 kg box 0.40 x 0.20 x 0.07; hips at x = +-0.18, y = +-0.12 (the L / R motors of a leg 0.02 to either side); upper legs
(motor rotor + link) 0.275 kg, length l1 chosen so that the reference's reset angles close the five-bar exactly (l1 = -0.2 cos 2.1834 = 0.1151);
lower legs 0.086 kg, 0.2 long; all 16 hinge axes parallel to the base's lateral (y) axis; contact spheres: 4 toes (r 0.012, on the R lower legs),
8 knees (r 0.015), 8 chassis corners (r 0.02); friction 1.0 (toes) / 0.5; MuJoCo-style soft constraints with default solref / solimp.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import physics_oracle as po  # noqa: E402  (model-compile time only, like tools/mjcf_compile.py)

LEGS = ('front_left', 'back_left', 'front_right', 'back_right')       # minitaur.py:17
MOTOR_DIR = np.array([-1, -1, -1, -1, 1, 1, 1, 1], float)             # minitaur.py:80
KNEE_ANGLE = -2.1834                                                  # minitaur.py:193
L2 = 0.2                                                              # minitaur.py:12-13 (constraint points at z = 0.2 of the knee links)
L1 = -L2 * np.cos(KNEE_ANGLE)                                         # closes the five-bar at the reset angles
DT, GRAVITY = 0.002, (0.0, 0.0, -10.0)
INIT_POSITION = (0.0, 0.0, 0.2)                                       # minitaur.py:10
NV = 22


def qy(angle):
  return np.array([np.cos(angle / 2), 0.0, np.sin(angle / 2), 0.0])


def box_inertia(m, hx, hy, hz):
  return np.array([m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy), 0, 0, 0])


def build():
  parent, jtype, tpos, tquat, jaxis, jpos = [], [], [], [], [], []
  mass, com, inertia, names = [], [], [], []

  def link(par, jt, tp=(0, 0, 0), tq=(1, 0, 0, 0), ax=(0, 0, 1), m=0.0, c=(0, 0, 0), I=(0, 0, 0, 0, 0, 0), name=''):
    parent.append(par); jtype.append(jt); tpos.append(tp); tquat.append(tq); jaxis.append(ax); jpos.append((0, 0, 0))
    mass.append(m); com.append(c); inertia.append(I); names.append(name)
    return len(parent) - 1
  # floating base: MuJoCo's free joint as six one-dof links (three slides along the world axes, then the orientation quaternion: types 2, 3, 3)
  for k in range(3):
    link(k - 1, 1, ax=np.eye(3)[k], name='base_t' + 'xyz'[k])
  link(2, 2, ax=(1, 0, 0), name='base_rx')
  link(3, 3, ax=(0, 1, 0), name='base_ry')
  base = link(4, 3, ax=(0, 0, 1), m=3.3, I=box_inertia(3.3, 0.20, 0.10, 0.035), name='base')
  motor_dofs, knee_dofs, lower_links, upper_links = [], [], {}, {}
  for g, leg in enumerate(LEGS):
    front, left = leg.startswith('front'), leg.endswith('left')
    hx, hy = (0.18 if front else -0.18), (0.12 if left else -0.12)
    for side in ('L', 'R'):
      # zero-angle direction of the upper leg (rotation about y that takes local +z there) and the sign of the knee's hinge axis: derived in
      # DESIGN.md section 14 from the reset angles and the leg model's extension / swing semantics (minitaur.py:187-211, 434-457)
      beta0 = np.pi if (left == (side == 'L')) else 0.0
      ksign = 1.0 if (left == (side == 'L')) else -1.0
      yh = hy + (0.02 if side == 'L' else -0.02)
      dy = -0.0275 if side == 'L' else 0.0175
      u = link(base, 0, tp=(hx, yh, 0.0), tq=qy(beta0), ax=(0, 1, 0), m=0.275, c=(0, 0, 0.015), I=(2.5e-4, 3.5e-4, 1.5e-4, 0, 0, 0),
               name=f'motor_{leg}{side}_joint')
      w = link(u, 0, tp=(0.0, dy, L1), ax=(0, ksign, 0), m=0.086, c=(0, 0, 0.1), I=(0.086 * L2 * L2 / 12, 0.086 * L2 * L2 / 12, 5e-6, 0, 0, 0),
               name=f'knee_{leg}{side}_link')
      motor_dofs.append(u); knee_dofs.append(w); upper_links[g, side] = u; lower_links[g, side] = w
  assert len(parent) == NV
  # attachments: base origin, then per leg the two ends of the loop closure (minitaur.py:12-13, 212-217: R knee link first)
  att_link, att_pos, att_names = [base], [(0, 0, 0)], ['base']
  con1, con2 = [], []
  for g, leg in enumerate(LEGS):
    att_link += [lower_links[g, 'R'], lower_links[g, 'L']]
    att_pos += [(0, 0.005, L2), (0, 0.01, L2)]
    att_names += [f'closure_{leg}R', f'closure_{leg}L']
    con1.append(len(att_link) - 2); con2.append(len(att_link) - 1)
  d = dict(parent=np.array(parent, np.int32), jtype=np.array(jtype, np.int32), tpos=np.array(tpos, float), tquat=np.array(tquat, float),
           jaxis=np.array(jaxis, float), jpos=np.array(jpos, float), mass=np.array(mass), com=np.array(com, float), inertia=np.array(inertia, float),
           att_link=np.array(att_link, np.int32), att_pos=np.array(att_pos, float), att_quat=np.tile([1.0, 0, 0, 0], (len(att_link), 1)),
           att_names=np.array(att_names), jnt_limited=np.zeros(NV, np.int32), jnt_range=np.zeros((NV, 2)), jnt_damping=np.zeros(NV),
           jnt_armature=np.zeros(NV), jnt_solref=np.tile([0.02, 1.0], (NV, 1)), jnt_solimp=np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (NV, 1)),
           dof_invweight0=np.ones(NV), act_joint=np.zeros(0, np.int32), act_kp=np.zeros(0), act_ctrlrange=np.zeros((0, 2)),
           weld_solref=np.array([[0.02, 1.0]]), weld_solimp=np.array([[0.9, 0.95, 0.001, 0.5, 2.0]]), weld_att=np.int32(-1),
           weld_invweight=np.ones(2), weld_mocap_quat=np.array([1.0, 0, 0, 0]), weld_mocap_pos=np.zeros(3),
           gravity=np.array(GRAVITY), timestep=np.float64(DT), ball_dof=np.int32(3),
           con_att1=np.array(con1, np.int32), con_att2=np.array(con2, np.int32), con_solref=np.tile([0.01, 1.0], (4, 1)),
           con_solimp=np.tile([0.95, 0.99, 0.001, 0.5, 2.0], (4, 1)), con_invweight=np.ones(4),
           motor_dof=np.array(motor_dofs, np.int32), knee_dof=np.array(knee_dofs, np.int32), motor_direction=MOTOR_DIR,
           link_names=np.array(names), leg_l1=np.float64(L1), leg_l2=np.float64(L2),
           # what the env randomizer reads back as "URDF masses" (minitaur.py:101-107: leg link LEG_LINK_ID[0], motor MOTOR_LINK_ID[0]): the two parts of this
           # model's 0.275 kg upper link (own authoring, like the link itself)
           rand_leg_mass=np.float64(0.034), rand_motor_mass=np.float64(0.241))
  # reset pose (minitaur.py:10-11, 187-211)
  q0 = np.zeros(NV + 1)
  q0[0:3], q0[3:7] = INIT_POSITION, (1, 0, 0, 0)
  for i, (u, w) in enumerate(zip(motor_dofs, knee_dofs)):
    q0[u + 1] = MOTOR_DIR[i] * np.pi / 2
    q0[w + 1] = MOTOR_DIR[i] * KNEE_ANGLE
  d['qpos0'] = q0
  # ---- collision geometry: spheres against the ground (a big box whose top face is z = 0) and the four walls
  sph_link, sph_pos, sph_r, sph_kind = [], [], [], []
  for g in range(4):                                  # toes: the end of the R lower leg (the L one is tied to it by the closure)
    sph_link.append(lower_links[g, 'R']); sph_pos.append((0, 0.005, L2)); sph_r.append(0.012); sph_kind.append(0)
  for g in range(4):                                  # knees: the origin of each lower leg
    for side in ('L', 'R'):
      sph_link.append(lower_links[g, side]); sph_pos.append((0, 0, 0)); sph_r.append(0.015); sph_kind.append(1)
  for sx in (-1, 1):                                  # chassis corners
    for sy in (-1, 1):
      for sz in (-1, 1):
        sph_link.append(base); sph_pos.append((0.19 * sx, 0.09 * sy, 0.025 * sz)); sph_r.append(0.02); sph_kind.append(2)
  box_pos = [(0, 0, -0.5)] + [(-1.55, 0, 0.5), (1.55, 0, 0.5), (0, -1.55, 0.5), (0, 1.55, 0.5)]
  box_half = [(10.0, 10.0, 0.5)] + [(0.05, 1.5, 0.25)] * 2 + [(1.5, 0.05, 0.25)] * 2
  nb = len(box_pos)
  pairs, pcls, blocks = [], [], []
  kinds = np.array(sph_kind)

  def block(box, sel, cap):
    b0 = len(pairs)
    for si in np.nonzero(sel)[0]:
      pairs.append((int(si), box)); pcls.append(int(kinds[si]))
    # bounding sphere of the set in the base frame: every sphere of the robot lies within 0.45 m of the base origin (leg reach 0.32 + hip offset)
    blocks.append(dict(begin=b0, end=len(pairs), box=box, link=base, center=(0, 0, 0), reach=0.5, cap=cap))
  block(0, kinds == 0, 4)                             # ground: toes first (the slot order is the priority order), then knees, then the chassis
  block(0, kinds == 1, 4)
  block(0, kinds == 2, 4)
  for b in range(1, nb):
    block(b, kinds == 2, 2)                           # walls: chassis corners (the tiles hang 0.25 m above the ground: legs pass under them)
  d.update(col_sph_link=np.array(sph_link, np.int32), col_sph_pos=np.array(sph_pos, float), col_sph_r=np.array(sph_r),
           col_sph_dir=np.zeros((len(sph_r), 3)), col_sph_hl=np.zeros(len(sph_r)),
           col_box_kind=np.zeros(nb, np.int32), col_box_link=np.full(nb, -1, np.int32), col_box_pos=np.array(box_pos, float),
           col_box_quat=np.tile([1.0, 0, 0, 0], (nb, 1)), col_box_half=np.array(box_half, float),
           col_pair=np.array(pairs, np.int32), col_pair_cls=np.array(pcls, np.int32),
           col_cls_mu=np.array([1.0, 0.5, 0.5]), col_cls_solref=np.tile([0.02, 1.0], (3, 1)),
           col_cls_solimp=np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (3, 1)), col_cls_margin=np.zeros(3), col_cls_invw=np.ones(3),
           col_blk_begin=np.array([b['begin'] for b in blocks], np.int32), col_blk_end=np.array([b['end'] for b in blocks], np.int32),
           col_blk_box=np.array([b['box'] for b in blocks], np.int32), col_blk_link=np.array([b['link'] for b in blocks], np.int32),
           col_blk_center=np.array([b['center'] for b in blocks], float), col_blk_reach=np.array([b['reach'] for b in blocks]),
           col_blk_cap=np.array([b['cap'] for b in blocks], np.int32), max_contacts=np.int32(12))
  # ---- inverse weights at the reset pose (MuJoCo's mj_setConst rule: mean diagonal of J M^-1 J' of the constrained points)
  lm = po.LinkModel(d)
  pos, quat, S = lm.kinematics(q0)
  out = lm.forward(q0, np.zeros(NV), np.zeros(0), np.zeros(3), np.array([1.0, 0, 0, 0]))
  Minv = np.linalg.inv(out['M'])

  def point_invweight(l, p):
    Jp = np.zeros((3, NV))
    for j in lm.anc[l]:
      Jp[:, j] = S[j, 3:] + np.cross(S[j, :3], p)
    return float(np.trace(Jp @ Minv @ Jp.T) / 3)
  inv = []
  for e in range(4):
    p1, _ = lm.attachment(pos, quat, con1[e]); p2, _ = lm.attachment(pos, quat, con2[e])
    assert np.abs(p1 - p2).max() < 1e-12, ('the reset pose closes the loops', p1, p2)
    inv.append(point_invweight(att_link[con1[e]], p1) + point_invweight(att_link[con2[e]], p2))
  d['con_invweight'] = np.array(inv)
  cls_inv = []
  for kcls in range(3):
    vals = [point_invweight(sph_link[si], pos[sph_link[si]] + po.quat_mat(quat[sph_link[si]]) @ np.array(sph_pos[si])) for si in np.nonzero(kinds == kcls)[0]]
    cls_inv.append(float(np.mean(vals)))
  d['col_cls_invw'] = np.array(cls_inv)
  d['dof_invweight0'] = np.diag(Minv).copy()
  return d


if __name__ == '__main__':
  d = build()
  out = os.path.join(ROOT, 'earl_benchmark_amd', 'models', 'minitaur_links.npz')
  np.savez_compressed(out, **d)
  lm = po.LinkModel(d)
  pos, quat, S = lm.kinematics(d['qpos0'])
  toes = [lm.attachment(pos, quat, k)[0] for k in d['con_att1']]
  print('wrote', out, 'nv', len(d['parent']), 'l1 %.6f' % L1, 'total mass %.3f' % d['mass'].sum())
  print('toe positions at reset:', np.round(np.array(toes), 4).tolist())
  print('con_invweight', d['con_invweight'], 'cls_invw', d['col_cls_invw'])
