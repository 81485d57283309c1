"""Batched articulated-body stepper (include/earl_physics.h): model tables -> device, raw step / forward calls.

STATUS (round 1): smooth dynamics, weld / joint-limit constraints and frictional contacts for the Sawyer-door model; parity with
MuJoCo unpinned.  The model tables are numeric facts compiled from the reference's MJCF by
tools/mjcf_compile.py into earl_benchmark_amd/models/*.npz.
"""
import ctypes as C
import os

import weakref

import numpy as np
import torch

from .. import _abi

MAXV, MAXATT, MAXACT = 16, 8, 4
MAXSPH, MAXBOX, MAXPAIR, MAXCLS, MAXCON, MAXBLK = 96, 16, 512, 16, 12, 64
MAXV24, MAXATT24, MAXJEQ, MAXCONNECT = 24, 16, 8, 4
MODEL_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'models')


class LinkModelStruct(C.Structure):   # struct earl_link_model
  _fields_ = [('nv', C.c_int32), ('n_att', C.c_int32), ('n_act', C.c_int32), ('weld_att', C.c_int32),
              ('n_jump', C.c_int32), ('ball_dof', C.c_int32), ('nq', C.c_int32), ('pad_', C.c_int32), ('jump', C.c_int32 * MAXV * 4),
              ('parent', C.c_int32 * MAXV), ('jtype', C.c_int32 * MAXV), ('limited', C.c_int32 * MAXV),
              ('anc_mask', C.c_uint32 * MAXV), ('desc_mask', C.c_uint32 * MAXV), ('att_link', C.c_int32 * MAXATT),
              ('act_joint', C.c_int32 * MAXACT),
              ('tpos', C.c_double * 3 * MAXV), ('tquat', C.c_double * 4 * MAXV), ('jaxis', C.c_double * 3 * MAXV),
              ('jpos', C.c_double * 3 * MAXV), ('mass', C.c_double * MAXV), ('com', C.c_double * 3 * MAXV),
              ('inertia', C.c_double * 6 * MAXV), ('range', C.c_double * 2 * MAXV), ('damping', C.c_double * MAXV),
              ('armature', C.c_double * MAXV), ('jsolref', C.c_double * 2 * MAXV), ('jsolimp', C.c_double * 5 * MAXV),
              ('dof_invweight', C.c_double * MAXV), ('att_pos', C.c_double * 3 * MAXATT), ('att_quat', C.c_double * 4 * MAXATT),
              ('act_kp', C.c_double * MAXACT), ('act_ctrlrange', C.c_double * 2 * MAXACT),
              ('weld_solref', C.c_double * 2), ('weld_solimp', C.c_double * 5), ('weld_invweight', C.c_double * 2),
              ('gravity', C.c_double * 3), ('dt', C.c_double), ('drag_G', C.c_double * MAXV), ('drag_b', C.c_double * MAXV),
              ('cd_mask', C.c_uint32 * MAXV)]


class LinkModelStruct24(C.Structure):   # struct earl_link_model24: up to 24 dofs + dry friction / springs / force limits / joint couplings
  _fields_ = [('nv', C.c_int32), ('n_att', C.c_int32), ('n_act', C.c_int32), ('weld_att', C.c_int32),
              ('n_jump', C.c_int32), ('ball_dof', C.c_int32), ('nq', C.c_int32), ('n_jeq', C.c_int32), ('jump', C.c_int32 * MAXV24 * 5),
              ('parent', C.c_int32 * MAXV24), ('jtype', C.c_int32 * MAXV24), ('limited', C.c_int32 * MAXV24),
              ('anc_mask', C.c_uint32 * MAXV24), ('desc_mask', C.c_uint32 * MAXV24), ('att_link', C.c_int32 * MAXATT24),
              ('act_joint', C.c_int32 * MAXACT), ('jeq_joint1', C.c_int32 * MAXJEQ), ('jeq_joint2', C.c_int32 * MAXJEQ),
              ('tpos', C.c_double * 3 * MAXV24), ('tquat', C.c_double * 4 * MAXV24), ('jaxis', C.c_double * 3 * MAXV24),
              ('jpos', C.c_double * 3 * MAXV24), ('mass', C.c_double * MAXV24), ('com', C.c_double * 3 * MAXV24),
              ('inertia', C.c_double * 6 * MAXV24), ('range', C.c_double * 2 * MAXV24), ('damping', C.c_double * MAXV24),
              ('armature', C.c_double * MAXV24), ('jsolref', C.c_double * 2 * MAXV24), ('jsolimp', C.c_double * 5 * MAXV24),
              ('dof_invweight', C.c_double * MAXV24), ('att_pos', C.c_double * 3 * MAXATT24), ('att_quat', C.c_double * 4 * MAXATT24),
              ('act_kp', C.c_double * MAXACT), ('act_ctrlrange', C.c_double * 2 * MAXACT),
              ('weld_solref', C.c_double * 2), ('weld_solimp', C.c_double * 5), ('weld_invweight', C.c_double * 2),
              ('gravity', C.c_double * 3), ('dt', C.c_double), ('drag_G', C.c_double * MAXV24), ('drag_b', C.c_double * MAXV24),
              ('cd_mask', C.c_uint32 * MAXV24), ('frictionloss', C.c_double * MAXV24), ('stiffness', C.c_double * MAXV24),
              ('springref', C.c_double * MAXV24), ('act_forcerange', C.c_double * 2 * MAXACT), ('jeq_coef', C.c_double * 2 * MAXJEQ),
              ('jeq_solref', C.c_double * 2 * MAXJEQ), ('jeq_solimp', C.c_double * 5 * MAXJEQ), ('jeq_invweight', C.c_double * MAXJEQ),
              ('pair', C.c_int32 * MAXV24),
              ('n_con', C.c_int32), ('con_att1', C.c_int32 * MAXCONNECT), ('con_att2', C.c_int32 * MAXCONNECT), ('pad3_', C.c_int32 * 3),
              ('con_solref', C.c_double * 2 * MAXCONNECT), ('con_solimp', C.c_double * 5 * MAXCONNECT), ('con_invweight', C.c_double * MAXCONNECT)]


class PairRec(C.Structure):
  _fields_ = [('sph_link', C.c_int32), ('cls', C.c_int32), ('pos', C.c_double * 3), ('r', C.c_double), ('margin', C.c_double),
              ('dir', C.c_double * 3), ('hl', C.c_double)]


class CollisionModelStruct(C.Structure):   # struct earl_collision_model
  _fields_ = [('n_sph', C.c_int32), ('n_box', C.c_int32), ('n_pair', C.c_int32), ('n_cls', C.c_int32),
              ('n_blk', C.c_int32), ('max_con', C.c_int32), ('cone', C.c_int32), ('pad_', C.c_int32), ('blk_begin', C.c_int32 * MAXBLK), ('blk_end', C.c_int32 * MAXBLK),
              ('blk_box', C.c_int32 * MAXBLK), ('blk_link', C.c_int32 * MAXBLK), ('blk_cap', C.c_int32 * MAXBLK), ('blk_center', C.c_double * 3 * MAXBLK),
              ('blk_reach', C.c_double * MAXBLK), ('blk_obb_center', C.c_double * 3 * MAXBLK), ('blk_obb_half', C.c_double * 3 * MAXBLK),
              ('sph_link', C.c_int32 * MAXSPH), ('box_link', C.c_int32 * MAXBOX),
              ('sph_pos', C.c_double * 3 * MAXSPH), ('sph_r', C.c_double * MAXSPH),
              ('box_pos', C.c_double * 3 * MAXBOX), ('box_quat', C.c_double * 4 * MAXBOX), ('box_half', C.c_double * 3 * MAXBOX),
              ('pair_sph', C.c_uint8 * MAXPAIR), ('pair_box', C.c_uint8 * MAXPAIR), ('pair_cls', C.c_uint8 * MAXPAIR),
              ('pair_kind', C.c_uint8 * MAXPAIR), ('pair_rec', PairRec * MAXPAIR),
              ('cls_mu', C.c_double * MAXCLS), ('cls_solref', C.c_double * 2 * MAXCLS), ('cls_solimp', C.c_double * 5 * MAXCLS),
              ('cls_margin', C.c_double * MAXCLS), ('cls_invw', C.c_double * MAXCLS), ('cls_mu_tor', C.c_double * MAXCLS)]


def _fill(dst, src):
  a = np.ctypeslib.as_array(dst)
  src = np.asarray(src)
  a[tuple(slice(0, k) for k in src.shape)] = src


def check_impedance_powers(d, keys, what):
  """csrc/physics_math.h imp_p2 -- the impedance of the kitchen's and the minitaur's kernels -- knows MuJoCo's solimp powers 1 and 2 (a constant impedance, d0 == dwidth,
  needs none): refuse any other here rather than compute a wrong impedance there"""
  for k in keys:
    if k in d and len(d[k]):
      a = np.asarray(d[k], float).reshape(-1, 5)
      ok = (a[:, 4] == 1) | (a[:, 4] == 2) | (a[:, 0] == a[:, 1])
      if not ok.all():
        raise _abi.EarlHipError(f'{what}: {k} row {int(np.flatnonzero(~ok)[0])} has solimp power {a[~ok][0, 4]:g}; the nv > 16 kernels implement powers 1 and 2 only')


def load_link_model(name):
  """-> (LinkModelStruct, dict of the npz arrays)"""
  with np.load(os.path.join(MODEL_DIR, name + '_links.npz')) as z:
    d = {k: z[k] for k in z.files}
  nv, natt, nact = len(d['parent']), len(d['att_link']), len(d['act_joint'])
  big = nv > MAXV                               # the kitchen: struct earl_link_model24
  assert (nv <= MAXV24 and natt <= MAXATT24 if big else nv <= MAXV and natt <= MAXATT) and nact <= MAXACT
  s = LinkModelStruct24() if big else LinkModelStruct()
  s.nv, s.n_att, s.n_act, s.weld_att = nv, natt, nact, int(d['weld_att'])
  s.ball_dof = int(d['ball_dof']) if 'ball_dof' in d else -1
  s.nq = nv + (1 if s.ball_dof >= 0 else 0)
  # a free body's rotation dofs are the last three (the peg), or the free body is the ROOT (the minitaur's base: dofs 0-5, MuJoCo's qpos layout)
  assert s.ball_dof < 0 or s.ball_dof == nv - 3 or (s.ball_dof == 3 and all(int(d['jtype'][k]) == t for k, t in enumerate((1, 1, 1, 2, 3, 3))))
  anc = np.zeros(nv, np.uint32)
  for l in range(nv):
    p = l
    while p >= 0:
      anc[l] |= np.uint32(1 << p)
      p = int(d['parent'][p])
  desc = np.array([sum(1 << i for i in range(nv) if (anc[i] >> l) & 1) for l in range(nv)], np.uint32)
  # links whose velocity enters d/dt of a link's axis: its ancestors; the three rotation axes of a free body all use the velocity
  # before any of them (oracle: LinkModel.forward, MuJoCo mj_comVel)
  cd = anc.copy()
  for l in range(nv):
    if int(d['jtype'][l]) == 3:
      pb = int(d['parent'][s.ball_dof])
      cd[l] = anc[pb] if pb >= 0 else 0
  # ancestor doubling tables for the log-depth kinematics
  par = [int(x) for x in d['parent']]
  depth = [bin(int(a)).count('1') for a in anc]
  s.n_jump = max(1, int(np.ceil(np.log2(max(depth)))))
  assert s.n_jump <= 4
  assert int(d['weld_att']) >= 0 or big, 'models without a mocap weld use the 24-dof table form'
  jump = np.full((4, MAXV24 if big else MAXV), -1, np.int32)
  for l in range(nv):
    chain = []
    p = par[l]
    while p >= 0:
      chain.append(p)
      p = par[p]
    for r in range(4):
      if len(chain) >= (1 << r):
        jump[r, l] = chain[(1 << r) - 1]
  _fill(s.jump, jump)
  for dst, src in ((s.parent, d['parent']), (s.jtype, d['jtype']), (s.limited, d['jnt_limited']), (s.anc_mask, anc),
                   (s.desc_mask, desc), (s.cd_mask, cd), (s.att_link, d['att_link']), (s.act_joint, d['act_joint']), (s.tpos, d['tpos']),
                   (s.tquat, d['tquat']), (s.jaxis, d['jaxis']), (s.jpos, d['jpos']), (s.mass, d['mass']), (s.com, d['com']),
                   (s.inertia, d['inertia']), (s.range, d['jnt_range']), (s.damping, d['jnt_damping']),
                   (s.armature, d['jnt_armature']), (s.jsolref, d['jnt_solref']), (s.jsolimp, d['jnt_solimp']),
                   (s.dof_invweight, d['dof_invweight0']), (s.att_pos, d['att_pos']), (s.att_quat, d['att_quat']),
                   (s.act_kp, d['act_kp']), (s.act_ctrlrange, d['act_ctrlrange']), (s.weld_solref, d['weld_solref'][0]),
                   (s.weld_solimp, d['weld_solimp'][0]), (s.weld_invweight, d['weld_invweight']), (s.gravity, d['gravity'])):
    _fill(dst, src)
  s.dt = float(d['timestep'])
  if 'dof_drag_G' in d:
    _fill(s.drag_G, d['dof_drag_G'])
    _fill(s.drag_b, d['dof_drag_b'])
  if big:
    check_impedance_powers(d, ('jnt_solimp', 'weld_solimp', 'jeq_solimp', 'con_solimp'), name)
  if big and nv == 22:
    # the minitaur: floating root + 16 hinges, no mocap weld, four connect constraints (csrc/physics.hip Lim<22>: dense factorisations)
    s.n_jeq, s.n_con = 0, len(d['con_att1'])
    assert s.n_con <= MAXCONNECT and s.weld_att < 0 and s.ball_dof == 3 and nact == 0
    for dst, src in ((s.con_att1, d['con_att1']), (s.con_att2, d['con_att2']), (s.con_solref, d['con_solref']), (s.con_solimp, d['con_solimp']),
                     (s.con_invweight, d['con_invweight'])):
      _fill(dst, src)
    _fill(s.pair, np.full(MAXV24, -1, np.int32))
    _fill(s.act_forcerange, np.tile([-1e300, 1e300], (MAXACT, 1)))
    # the tree the minitaur's own timestep is written on (csrc/minitaur_stepper.h): a root body of six one-dof links with identity frames, then four legs
    # of four hinges -- leg k = dofs 6 + 4 k + (0: upper link of the first chain, 1: its lower link, 2: upper link of the second chain, 3: its lower
    # link), hinge frames without a joint offset; closure k ties attachment con_att1[k] on hinge 3 of leg k to con_att2[k] on its hinge 1; no joint
    # limits, no damping, no dragging contact (their rows are not built there)
    par, jt = [int(x) for x in d['parent']], [int(x) for x in d['jtype']]
    assert par[:6] == [-1, 0, 1, 2, 3, 4] and jt[:6] == [1, 1, 1, 2, 3, 3], 'minitaur tree: the root body'
    assert np.array_equal(d['jaxis'][:6], np.vstack([np.eye(3), np.eye(3)])) and not np.any(d['tpos'][:6]) and np.array_equal(d['tquat'][:6], np.tile([1.0, 0, 0, 0], (6, 1)))
    assert all(par[6 + 4 * k + j] == (5, 6 + 4 * k, 5, 8 + 4 * k)[j] and jt[6 + 4 * k + j] == 0 for k in range(4) for j in range(4)), 'minitaur tree: the legs'
    assert not np.any(d['jpos']) and not np.any(d['jnt_limited']) and not np.any(d['jnt_damping']) and 'dof_drag_G' not in d
    assert s.n_con == 4 and all(int(d['att_link'][int(d['con_att1'][k])]) == 9 + 4 * k and int(d['att_link'][int(d['con_att2'][k])]) == 7 + 4 * k for k in range(4)), \
        'minitaur tree: closure k between hinge 3 and hinge 1 of leg k'
  elif big:
    s.n_jeq = len(d['jeq_joint1'])
    assert s.n_jeq <= MAXJEQ
    fr = np.array(d['act_forcerange'], float)
    fr = np.where(np.isfinite(fr), fr, np.sign(fr) * 1e300)          # +-inf = not force limited
    for dst, src in ((s.frictionloss, d['jnt_frictionloss']), (s.stiffness, d['jnt_stiffness']), (s.springref, d['jnt_springref']),
                     (s.act_forcerange, fr), (s.jeq_joint1, d['jeq_joint1']), (s.jeq_joint2, d['jeq_joint2']), (s.jeq_coef, d['jeq_coef']),
                     (s.jeq_solref, d['jeq_solref']), (s.jeq_solimp, d['jeq_solimp']), (s.jeq_invweight, d['jeq_invweight'])):
      _fill(dst, src)
    # structure the kernel's solver relies on: the first 9 dofs are one tree (the arm), every other dof is its own tree, couplings tie
    # fixtures in disjoint pairs
    pair = np.full(MAXV24, -1, np.int32)
    for j1, j2 in zip(d['jeq_joint1'], d['jeq_joint2']):
      assert pair[j1] < 0 and pair[j2] < 0 and j1 >= 9 and j2 >= 9
      pair[j1], pair[j2] = j2, j1
    assert nv == 23 and all(int(d['parent'][l]) == (l - 1 if 0 < l < 7 else (6 if l in (7, 8) else -1)) for l in range(nv)), 'kitchen tree shape'
    assert not np.any(d['jpos']), 'csrc/physics.hip Lim<23>::ARMSCAN: frames are composed as tpos / tquat o joint (no joint offsets)'
    _fill(s.pair, pair)
  else:
    assert not np.any(d.get('jnt_frictionloss', 0)) and not np.any(d.get('jnt_stiffness', 0)) and len(d.get('jeq_joint1', ())) == 0, \
        'dry friction / springs / joint couplings need the 24-dof model form'
    if nv == 15:
      # csrc/physics.hip Lim<15>::TS = 9: two trees, links [0, 9) and [9, 15), with the weld on the first -- the lanes' ancestor / subtree sums and the
      # contact rows are restricted to one tree each on the strength of this
      par = [int(x) for x in d['parent']]
      root = lambda l: l if par[l] < 0 else root(par[l])
      assert all(root(l) == 0 for l in range(9)) and all(root(l) == 9 for l in range(9, 15)), 'peg model: trees [0, 9) and [9, 15)'
      assert int(d['att_link'][int(d['weld_att'])]) < 9, 'peg model: the weld sits on the first tree'
      # csrc/physics.hip Lim<15>::ARMSCAN: the scans along the chains assume the arm = seven hinges in series (links 0-6) + the two fingers on the hand, and
      # the free body = links 9-14 in series: three slides along the world axes from the origin, the quaternion link, two rigid links; the mass on the last
      assert par == [-1, 0, 1, 2, 3, 4, 5, 6, 6, -1, 9, 10, 11, 12, 13] and [int(x) for x in d['jtype']][9:] == [1, 1, 1, 2, 3, 3] and int(d['ball_dof']) == 12
      assert np.array_equal(d['jaxis'][9:12], np.eye(3)) and not np.any(d['tpos'][9:]) and np.array_equal(d['tquat'][9:], np.tile([1.0, 0, 0, 0], (6, 1)))
      assert not np.any(d['jpos']) and not np.any(d['mass'][9:14])
  return s, d


def load_collision_model(d):
  """tables dict (from load_link_model) -> CollisionModelStruct, or None if the model has no collision geometry"""
  if 'col_pair' not in d or len(d['col_pair']) == 0:
    return None
  c = CollisionModelStruct()
  c.n_sph, c.n_box, c.n_pair, c.n_cls = len(d['col_sph_link']), len(d['col_box_link']), len(d['col_pair']), len(d['col_cls_mu'])
  c.n_blk = len(d['col_blk_begin'])
  c.max_con = int(d['max_contacts']) if 'max_contacts' in d else 8
  c.cone = int(d['cone_elliptic']) if 'cone_elliptic' in d else 0
  assert c.cone == (1 if len(d['parent']) <= 16 else 0), 'csrc/physics.hip Lim<NV>::ELLIPTIC: the Sawyer models (nv 10, 15) are compiled with the elliptic cone, the others with the pyramid'
  small = len(d['parent']) <= 10                # csrc/physics.hip Lim<NV>: 8 contact slots / 16 blocks for nv <= 10
  if len(d['parent']) > 16:
    check_impedance_powers(d, ('col_cls_solimp',), 'collision classes')
  assert c.n_sph <= MAXSPH and c.n_box <= MAXBOX and c.n_pair <= MAXPAIR and c.n_cls <= MAXCLS
  nvm = len(d['parent'])
  assert c.n_blk <= (16 if small else (64 if nvm == 23 else (8 if nvm == 22 else 32))) and 0 < c.max_con <= (8 if small else MAXCON)      # Lim<NV>::MB, ::MC
  if nvm == 23:      # csrc/physics.hip Lim<23>::PACK: several near blocks share a pass of the group's 32 lanes; a block must fit one pass
    assert int(np.max(np.asarray(d['col_blk_end']) - np.asarray(d['col_blk_begin']))) <= 32, 'kitchen collision blocks: at most 32 pairs each'
  if nvm == 22:
    # csrc/physics.hip (Lim<22>::CONNECT, K9): a contact's Jacobian is taken to touch the root body's six dofs and the sphere's own chain of at most
    # two hinges -- spheres ride on the root body or on such a chain, boxes are fixed to the world
    par = [int(x) for x in d['parent']]
    assert all(int(b) < 0 for b in d['col_box_link']), 'minitaur collision model: world-fixed boxes only'
    assert all(int(l) == 5 or (int(l) >= 6 and (par[int(l)] == 5 or (par[int(l)] >= 6 and par[par[int(l)]] == 5))) for l in d['col_sph_link']), 'sphere links: root body or a chain of <= 2 hinges'
  for dst, src in ((c.blk_begin, d['col_blk_begin']), (c.blk_end, d['col_blk_end']), (c.blk_box, d['col_blk_box']),
                   (c.blk_link, d['col_blk_link']), (c.blk_center, d['col_blk_center']), (c.blk_reach, d['col_blk_reach']),
                   (c.blk_cap, d['col_blk_cap'] if 'col_blk_cap' in d else np.full(c.n_blk, c.max_con, np.int32)),
                   (c.blk_obb_center, d['col_blk_obb_center'] if 'col_blk_obb_half' in d else np.zeros((c.n_blk, 3))),
                   (c.blk_obb_half, d['col_blk_obb_half'] if 'col_blk_obb_half' in d else np.full((c.n_blk, 3), 1e30))):   # (absent: never separated)
    _fill(dst, src)
  for dst, src in ((c.sph_link, d['col_sph_link']), (c.box_link, d['col_box_link']), (c.sph_pos, d['col_sph_pos']), (c.sph_r, d['col_sph_r']),
                   (c.box_pos, d['col_box_pos']), (c.box_quat, d['col_box_quat']), (c.box_half, d['col_box_half']),
                   (c.pair_sph, d['col_pair'][:, 0].astype(np.uint8)), (c.pair_box, d['col_pair'][:, 1].astype(np.uint8)),
                   (c.pair_cls, d['col_pair_cls'].astype(np.uint8)), (c.cls_mu, d['col_cls_mu']), (c.cls_solref, d['col_cls_solref']),
                   (c.cls_solimp, d['col_cls_solimp']), (c.cls_margin, d['col_cls_margin']), (c.cls_invw, d['col_cls_invw'])):
    _fill(dst, src)
  for i, (si, bi) in enumerate(d['col_pair']):
    r = c.pair_rec[i]
    r.sph_link, r.cls, r.r = int(d['col_sph_link'][si]), int(d['col_pair_cls'][i]), float(d['col_sph_r'][si])
    r.pos[:] = [float(x) for x in d['col_sph_pos'][si]]
    r.margin = float(d['col_cls_margin'][r.cls])
    if 'col_sph_dir' in d:                      # edges (segments) of the edge-vs-capsule blocks; zero for spheres / points
      r.dir[:] = [float(x) for x in d['col_sph_dir'][si]]
      r.hl = float(d['col_sph_hl'][si])
  if 'col_cls_mu_tor' in d:                     # condim-4 classes (round 5): torsional coefficient, 0 = none
    _fill(c.cls_mu_tor, d['col_cls_mu_tor'])
    assert c.cone == 1 or not np.any(d['col_cls_mu_tor']), 'torsional rows exist for the elliptic cone only'
  if 'col_pair_kind' in d:                      # 2 = cylinder vs box (round 5); 0 / 1 as the block says
    _fill(c.pair_kind, np.asarray(d['col_pair_kind']).astype(np.uint8))
  if 'col_box_kind' in d:                       # kind of each block = kind of its box (1: capsule), carried in bit 8 of blk_cap
    for b in range(c.n_blk):
      assert 0 < c.blk_cap[b] < 256
      kind = int(d['col_box_kind'][int(d['col_blk_box'][b])])
      assert kind == 0 or small, 'edge-vs-capsule blocks are compiled in for nv <= 10 only (csrc/physics.hip Lim<NV>::CAPS)'
      c.blk_cap[b] |= kind << 8
  return c


def check_layouts(lib):
  """the hand-mirrored ctypes layouts are memcpy'd to the device: refuse a library compiled against other struct layouts"""
  for fn, mirror in ((lib.earl_physics_model_size, LinkModelStruct), (lib.earl_physics_model24_size, LinkModelStruct24),
                     (lib.earl_collision_model_size, CollisionModelStruct),
                     (lib.earl_sawyer_cfg_size, _abi.SawyerCfg), (lib.earl_minitaur_cfg_size, _abi.MinitaurCfg)):
    if fn() != C.sizeof(mirror):
      raise _abi.EarlHipError(f'{mirror.__name__}: the library was built with sizeof = {fn()}, this binding expects {C.sizeof(mirror)} '
                              '(stale csrc/libearl_hip.so or header drift: rebuild with __graft_entry__.build())')


class DeviceModel:
  """a link model resident in HBM"""

  def __init__(self, name, device='cuda', contacts=True):
    self.lib = _abi.load()
    check_layouts(self.lib)
    self.struct, self.tables = load_link_model(name)
    self.col_struct = load_collision_model(self.tables) if contacts else None
    if self.col_struct is not None and (any(self.col_struct.pair_kind[i] == 2 for i in range(self.col_struct.n_pair)) or any(self.col_struct.cls_mu_tor[:])):
      # include/earl_physics.h: cylinder-vs-box pairs (pair_kind 2) and torsional rows (cls_mu_tor) exist in the C restatement's experimental tables only;
      # the kernels would read a cylinder record as a sphere of the cylinder's radius
      raise _abi.EarlHipError(f'{name}: this collision table uses cylinder pairs / torsional rows, which the HIP kernels do not implement (experimental tables of the restatement)')
    if not contacts:
      _fill(self.struct.drag_G, np.zeros(self.struct.nv))
    self.nv, self.n_att, self.n_act, self.nq = self.struct.nv, self.struct.n_att, self.struct.n_act, self.struct.nq
    self.device = torch.device(device)
    raw = np.frombuffer(bytes(self.struct), dtype=np.uint8).copy()
    self.buf = torch.from_numpy(raw).to(self.device)
    self.col_buf = None
    if self.col_struct is not None:
      self.col_buf = torch.from_numpy(np.frombuffer(bytes(self.col_struct), dtype=np.uint8).copy()).to(self.device)
      # the library remembers a table's friction cone per device address (include/earl_physics.h: earl_physics_forget_table): this block may have been another
      # table's before (torch's caching allocator), and will be again after this model is gone
      self.lib.earl_physics_forget_table(self.col_buf.data_ptr())
      weakref.finalize(self, self.lib.earl_physics_forget_table, self.col_buf.data_ptr())
    self.att_names = [str(x) for x in self.tables['att_names']]

  def _stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  @property
  def col_ptr(self):
    return None if self.col_buf is None else self.col_buf.data_ptr()

  def step(self, qpos, qvel, mocap_pos, mocap_quat, ctrl, nsub=1, att_xpos=None):
    n = qpos.shape[0]
    with torch.cuda.device(self.device):
      rc = self.lib.earl_physics_step(self.buf.data_ptr(), self.col_ptr, self.nv, n, nsub, qpos.data_ptr(), qvel.data_ptr(), mocap_pos.data_ptr(),
                                      mocap_quat.data_ptr(), ctrl.data_ptr(), None if att_xpos is None else att_xpos.data_ptr(),
                                      self._stream())
    _abi.check(rc, 'physics_step')

  def forward(self, qpos, qvel, mocap_pos, mocap_quat, ctrl):
    n = qpos.shape[0]
    kw = dict(dtype=torch.float64, device=self.device)
    qacc = torch.empty(n, self.nv, **kw)
    efc = torch.empty(n, 6 + 2 * self.nv, **kw)
    att = torch.empty(n, self.n_att, 3, **kw)
    with torch.cuda.device(self.device):
      rc = self.lib.earl_physics_forward(self.buf.data_ptr(), self.col_ptr, self.nv, n, qpos.data_ptr(), qvel.data_ptr(), mocap_pos.data_ptr(),
                                         mocap_quat.data_ptr(), ctrl.data_ptr(), qacc.data_ptr(), efc.data_ptr(), att.data_ptr(),
                                         self._stream())
    _abi.check(rc, 'physics_forward')
    return qacc, efc, att
