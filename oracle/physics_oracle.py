"""CPU reference of the articulated-body stepper (numpy, fp64) for the models compiled by tools/mjcf_compile.py.

TEST INFRASTRUCTURE, NOT PRODUCT.  It is also used by tools/mjcf_compile.py at model-compile time (inverse weights at
qpos0).  **Parity with MuJoCo is UNPINNED**: MuJoCo 2.1 / metaworld are not in /root/reference and cannot be run here
(SURVEY.md section 8c).  The algorithm follows MuJoCo's documented pipeline (Computation chapter):

  kinematics -> composite-rigid-body mass matrix (+ armature) -> RNE bias forces (gravity, Coriolis) -> passive joint
  damping -> position actuators -> soft constraints (weld to the mocap body, joint limits; contacts: NOT YET) with
  solref / solimp impedance and the dual problem (A + R) f = aref - J a0 solved by projected Gauss-Seidel ->
  semi-implicit Euler with implicit joint damping.

(That paragraph describes the body-level functions at the top of this file.  The reduced LinkModel further down -- the statement the HIP kernels follow -- adds the
frictional contacts, with the cone the model's MJCF declares: four pyramid edges per contact, or MuJoCo's elliptic three-zone cost, solve_primal_elliptic, for the Sawyer
scenes since round 4; its constraint solve is the primal active-set Newton iteration the kernels run.)

What IS pinned by the reference's own data: since round 4 the arm + weld dynamics on the contact-free prefixes of the 40 recorded Sawyer episodes (hand path within 1 mm RMS,
tools/weld_free_motion_fit.py; two weld factors calibrated there, declared below); and the model tables + forward kinematics reproduce the two door-handle
positions recorded in envs/sawyer_door.py:46-47 to 1e-8 (tests/test_physics.py).
All joints are 1-dof (hinge / slide), so nq == nv.
"""
import numpy as np


# ------------------------------------------------------------------ quaternions / rotations
def quat_mul(a, b):
  w1, x1, y1, z1 = a
  w2, x2, y2, z2 = b
  return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                   w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quat_conj(q):
  return np.array([q[0], -q[1], -q[2], -q[3]])


# Regularisers of the mocap weld's rows as multiples of the values this file derives from the MJCF
# (R = (1 - d) / d * body_invweight0[hand], translational / rotational part).  CALIBRATIONS against the MuJoCo recordings, declared as such.
#
# Round 4 (tools/weld_free_motion_fit.py -> profiles/r04_weld_free_motion_fit.json): both factors are identified on the
# CONTACT-FREE prefixes of the recorded Sawyer episodes (door 13 / 38 env steps, peg 11), where the hand path depends only on the arm's
# MJCF constants, the weld and the state the episode starts from.  Least squares over the start state (one per task) and the two factors,
# FIT-set episodes (even-numbered) only, door and peg jointly: translation x 3.35, rotation x 0.07; replayed hand path within 0.95 / 0.81 mm
# RMS of the recorded one (held-out prefixes 1.01 / 0.92 mm; with round 2's (4.0, 1.0): 4.8 / 1.6 mm even with the start state free).  The
# two tasks fitted separately agree (door 3.2 / 0.06, peg 3.6 / 0.08); freeing the arm's joint damping as well leaves it within 10 % of the MJCF
# value, i.e. the arm model is not what absorbs the factors.  What the rotational factor means physically: the wrist's three weld rows are
# 14 x stiffer than derived, the hand tilts 2 - 3 degrees under the loads of the door pull instead of round 3's 18.
# Which MuJoCo 2.1 rule produces either factor is not identified (3.35 is close to what body_invweight0 becomes if J M^-1 J^T is replaced
# by J diag(dof_invweight0) J^T at qpos0, 3.52 -- but the same rule applied to the contact rows makes every replay worse, so it is not adopted).
# Round 1 - 3 values, for the record: translation 4.0 (read off the hand's sag at reset, which round 4 found to be a transient, not a static
# quantity), rotation 0.5 (round 1, fitted) = "mocap quaternion unnormalised" (round 2, a rule with the same effect; the quaternion is still
# taken as given, the factor below is on top of it).
# Models without recordings (the kitchen) keep the round-2 values: tools/mjcf_compile.py passes them explicitly.
WELD_TRANSLATION_CALIBRATION = 3.35
WELD_ROTATION_CALIBRATION = 0.07
LEGACY_WELD_CALIBRATION = (4.0, 1.0)          # rounds 2 - 3; what a model file without the key 'weld_calibration' was compiled with


def quat_mat(q):
  w, x, y, z = q
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def skew(v):
  return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


class Model:
  """tables of one compiled model (earl_benchmark_amd/models/<name>.npz)"""

  def __init__(self, path_or_dict):
    d = np.load(path_or_dict, allow_pickle=False) if isinstance(path_or_dict, str) else path_or_dict
    for k in d.files if hasattr(d, 'files') else d:
      setattr(self, k, np.array(d[k]))
    self.nb, self.nv = len(self.body_parent), len(self.jnt_body)
    self.dt = float(self.timestep)
    if not hasattr(self, 'weld_calibration'):
      self.weld_calibration = np.array(LEGACY_WELD_CALIBRATION)
    # ancestors of each dof (dofs whose motion moves the dof's body), incl. itself
    self.body_dofs = [[] for _ in range(self.nb)]
    for j in range(self.nv):
      self.body_dofs[self.jnt_body[j]].append(j)
    self.dof_anc = []                      # dofs that move dof j's body: those of its ancestors and, on its own body, j and earlier
    for j in range(self.nv):
      anc, b = [a for a in self.body_dofs[self.jnt_body[j]] if a <= j], self.body_parent[self.jnt_body[j]]
      while b != 0:
        anc = self.body_dofs[b] + anc
        b = self.body_parent[b]
      self.dof_anc.append(anc)

  def body_id(self, name):
    return list(self.body_names).index(name)

  def geom_id(self, name):
    return list(self.geom_names).index(name)

  def site_id(self, name):
    return list(self.site_names).index(name)


# ------------------------------------------------------------------ kinematics
def kinematics(m, qpos, body_pos=None):
  """-> dict(xpos [nb,3], xquat [nb,4], xmat [nb,3,3], xipos, ximat, axis [nv,3] world, anchor [nv,3] world)"""
  bp = m.body_pos if body_pos is None else body_pos
  xpos, xquat = np.zeros((m.nb, 3)), np.tile([1.0, 0, 0, 0], (m.nb, 1))
  axis, anchor = np.zeros((m.nv, 3)), np.zeros((m.nv, 3))
  for b in range(1, m.nb):
    p = m.body_parent[b]
    R = quat_mat(xquat[p])
    xpos[b] = xpos[p] + R @ bp[b]
    xquat[b] = quat_mul(xquat[p], m.body_quat[b])
    for j in m.body_dofs[b]:
      Rb = quat_mat(xquat[b])
      anchor[j] = xpos[b] + Rb @ m.jnt_pos[j]
      axis[j] = Rb @ m.jnt_axis[j]
      if m.jnt_type[j] != 1:   # hinge: rotate about the axis through the anchor (the components of a free joint's rotation
                               # count as hinges here: exact at zero rotation, which is all this full-body form is used at)
        h = 0.5 * qpos[j]
        xquat[b] = quat_mul(xquat[b], np.concatenate([[np.cos(h)], np.sin(h) * m.jnt_axis[j]]))
        xpos[b] = anchor[j] - quat_mat(xquat[b]) @ m.jnt_pos[j]
      else:                    # slide
        xpos[b] = xpos[b] + axis[j] * qpos[j]
  xmat = np.stack([quat_mat(q) for q in xquat])
  xipos = xpos + np.einsum('bij,bj->bi', xmat, m.body_ipos)
  ximat = np.stack([xmat[b] @ quat_mat(m.body_iquat[b]) for b in range(m.nb)])
  return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat, axis=axis, anchor=anchor)


def motion_subspace(m, kin):
  """S [nv,6]: spatial velocity (omega, v of the point at the world origin) per unit joint velocity"""
  S = np.zeros((m.nv, 6))
  for j in range(m.nv):
    if m.jnt_type[j] != 1:
      S[j, :3] = kin['axis'][j]
      S[j, 3:] = np.cross(kin['anchor'][j], kin['axis'][j])
    else:
      S[j, 3:] = kin['axis'][j]
  return S


def spatial_inertia(mass, c, Ic):
  """6x6 inertia about the world origin of a body with COM c and rotational inertia Ic (world axes)"""
  cx = skew(c)
  I = np.zeros((6, 6))
  I[:3, :3] = Ic + mass * cx @ cx.T
  I[:3, 3:] = mass * cx
  I[3:, :3] = mass * cx.T
  I[3:, 3:] = mass * np.eye(3)
  return I


def crossm(v):  # spatial motion cross product  v x .
  w, u = v[:3], v[3:]
  X = np.zeros((6, 6))
  X[:3, :3] = skew(w); X[3:, :3] = skew(u); X[3:, 3:] = skew(w)
  return X


def crossf(v):  # spatial force cross product  v x* .
  return -crossm(v).T


def body_inertias(m, kin):
  return [spatial_inertia(m.body_mass[b], kin['xipos'][b], kin['ximat'][b] @ np.diag(m.body_inertia[b]) @ kin['ximat'][b].T)
          for b in range(m.nb)]


def mass_matrix(m, kin, S=None, I=None):
  S = motion_subspace(m, kin) if S is None else S
  Ic = [x.copy() for x in (body_inertias(m, kin) if I is None else I)]
  for b in range(m.nb - 1, 0, -1):
    Ic[m.body_parent[b]] += Ic[b]
  M = np.zeros((m.nv, m.nv))
  for i in range(m.nv):
    F = Ic[m.jnt_body[i]] @ S[i]
    for j in m.dof_anc[i]:
      M[i, j] = M[j, i] = S[j] @ F
  M[np.arange(m.nv), np.arange(m.nv)] += m.jnt_armature
  return M


def bias_forces(m, kin, qvel, S=None, I=None):
  """RNE with zero acceleration: Coriolis / centrifugal + gravity generalized forces (qfrc_bias)"""
  S = motion_subspace(m, kin) if S is None else S
  I = body_inertias(m, kin) if I is None else I
  V = np.zeros((m.nb, 6)); A = np.zeros((m.nb, 6))
  A[0, 3:] = -m.gravity                   # fictitious base acceleration
  F = np.zeros((m.nb, 6))
  for b in range(1, m.nb):
    p = m.body_parent[b]
    V[b], A[b] = V[p], A[p].copy()
    for j in m.body_dofs[b]:
      A[b] = A[b] + crossm(V[b]) @ S[j] * qvel[j]
      V[b] = V[b] + S[j] * qvel[j]
    F[b] = I[b] @ A[b] + crossf(V[b]) @ (I[b] @ V[b])
  for b in range(m.nb - 1, 0, -1):
    F[m.body_parent[b]] += F[b]
  return np.array([S[j] @ F[m.jnt_body[j]] for j in range(m.nv)])


def body_jacobian(m, S, b, point):
  """6 x nv: (omega, velocity of `point` rigidly attached to body b)"""
  J = np.zeros((6, m.nv))
  bb = b
  while bb != 0:
    for j in m.body_dofs[bb]:
      J[:3, j] = S[j, :3]
      J[3:, j] = S[j, 3:] + np.cross(S[j, :3], point)
    bb = m.body_parent[bb]
  return J


# ------------------------------------------------------------------ soft constraints (MuJoCo solref / solimp)
def impedance(solimp, r):
  d0, dw, width, mid, power = solimp
  x = min(abs(r) / width, 1.0) if width > 0 else 1.0
  if power == 1 or d0 == dw:
    y = x
  elif x <= mid:
    y = x ** power / mid ** (power - 1)
  else:
    y = 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
  return d0 + y * (dw - d0)


def kbimp(solref, solimp, r, dt):
  tc, dr = max(solref[0], 2 * dt), solref[1]
  dmax = solimp[1]
  d = impedance(solimp, r)
  k = 1.0 / (dmax * dmax * tc * tc * dr * dr)
  b = 2.0 / (dmax * tc)
  return k, b, d


WELD_FORM = 'shipped'      # 'shipped' | 'documented' (tools/weld_free_motion_fit.py --forms evaluates both, and six more, on the contact-free prefixes)


def constraints(m, kin, S, qpos, qvel, mocap_pos, mocap_quat, weld_form=None):
  """rows: J [nc,nv], r [nc], aref [nc], R [nc], is_equality [nc]

  The mocap weld's six rows in two FORMS (weld_form, default WELD_FORM):
  * 'documented' -- MuJoCo's weld as its documentation and mj_instantiateEquality describe it: the mocap body's quaternion NORMALISED (mj_kinematics normalises mocap_quat),
    translational residual p_mocap - p_hand with the point Jacobian; rotational residual = vector part of the difference quaternion conj(q_hand) q_mocap (~ half the rotation
    vector) with Jacobian 0.5 x (angular Jacobian in the hand's axes, corrected by the quaternion); regulariser R = (1 - d) / d x body_invweight0[hand] -- its translational
    component on the first three rows, its rotational one on the last three (diagApprox of an equality row) -- with d the row's own impedance.  Factor 1.0 on everything.
  * 'shipped' -- the same rows with the mocap quaternion taken AS GIVEN ([1, 0, 1, 0], norm sqrt 2: both the residual and the Jacobian of the rotational rows come out
    sqrt 2 larger) and the two regularisers multiplied by WELD_TRANSLATION_CALIBRATION / WELD_ROTATION_CALIBRATION (m.weld_calibration).
  On the contact-free prefixes of the recordings (profiles/r05_weld_forms.json; start state re-fitted per form): 'documented' misses the recorded hand path by 13.4 mm (door) /
  4.9 mm (peg) RMS, 'shipped' by 0.94 / 0.77 mm.  No documented form needs no factor."""
  form = weld_form or WELD_FORM
  rows = []
  for w in range(len(m.weld_body1)):
    b1, b2 = int(m.weld_body1[w]), int(m.weld_body2[w])
    assert m.body_mocap[b1], 'weld body1 is the mocap body in these models'
    p1, q1 = mocap_pos, np.asarray(mocap_quat, dtype=np.float64)       # as given: NOT normalised (round 2's rule, see the note above WELD_TRANSLATION_CALIBRATION)
    cal = np.asarray(m.weld_calibration, float)
    if form == 'documented':
      q1, cal = q1 / np.sqrt(q1 @ q1), np.ones(2)
    p2, q2 = kin['xpos'][b2], kin['xquat'][b2]
    Jb = body_jacobian(m, S, b2, p2)
    # mj_instantiateEqual, weld with relpose = identity (metaworld's reset_mocap_welds): position error body1 - body2;
    # orientation error = vec(conj(q2) * q1) with its exact Jacobian -0.5 (e_w a + a x e_v), a = R2^T w_j
    rpos = p1 - p2
    e = quat_mul(quat_conj(q2), q1)
    R2 = quat_mat(q2)
    for a in range(3):
      rows.append((-Jb[3 + a], rpos[a], m.weld_solref[w], m.weld_solimp[w], float(cal[0]) * m.body_invweight0[b2, 0], True))
    A = R2.T @ Jb[0:3]                                        # 3 x nv, angular Jacobian in body2 axes
    Jq = -0.5 * (e[0] * A + np.cross(A.T, e[1:]).T)
    for a in range(3):
      rows.append((Jq[a], e[1 + a], m.weld_solref[w], m.weld_solimp[w], float(cal[1]) * m.body_invweight0[b2, 1], True))
  for j in range(m.nv):
    if m.jnt_limited[j]:
      lo, hi = m.jnt_range[j]
      e = np.zeros(m.nv); e[j] = 1.0
      if qpos[j] - lo < 0:
        rows.append((e, qpos[j] - lo, m.jnt_solref[j], m.jnt_solimp[j], m.dof_invweight0[j], False))
      if hi - qpos[j] < 0:
        rows.append((-e, hi - qpos[j], m.jnt_solref[j], m.jnt_solimp[j], m.dof_invweight0[j], False))
  if not rows:
    return np.zeros((0, m.nv)), np.zeros(0), np.zeros(0), np.zeros(0), np.zeros(0, bool)
  J = np.stack([r[0] for r in rows])
  res = np.array([r[1] for r in rows])
  aref, R = np.zeros(len(rows)), np.zeros(len(rows))
  for i, (Ji, ri, solref, solimp, invw, _) in enumerate(rows):
    k, b, d = kbimp(solref, solimp, ri, m.dt)
    aref[i] = -b * (Ji @ qvel) - k * d * ri
    R[i] = max((1 - d) / d * invw, 1e-15)
  return J, res, aref, R, np.array([r[5] for r in rows])


def solve_constraints(A, R, rhs, is_eq, iters=5000):
  """(A + R) f = rhs with f >= 0 on the inequality rows: projected Gauss-Seidel, swept until the forces stop changing (round 4: the rotational weld rows carry a
  14 x smaller regulariser, with which the 50 sweeps of rounds 1 - 3 are not enough)"""
  n = len(rhs)
  f = np.zeros(n)
  AR = A + np.diag(R)
  for _ in range(iters):
    change = 0.0
    for i in range(n):
      fi = f[i] + (rhs[i] - AR[i] @ f) / AR[i, i]
      if not is_eq[i] and fi < 0:
        fi = 0.0
      change = max(change, abs(fi - f[i]))
      f[i] = fi
    if change <= 1e-12 * (1.0 + np.abs(f).max()):
      break
  return f


class State:
  def __init__(self, m):
    self.qpos = np.zeros(m.nv); self.qvel = np.zeros(m.nv)
    self.mocap_pos = np.zeros(3); self.mocap_quat = np.array([1.0, 0, 0, 0])
    self.ctrl = np.zeros(len(m.act_joint))


def forward(m, s, body_pos=None):
  """accelerations of one state -> dict with qacc (before the integrator's implicit damping), M, forces, kinematics"""
  kin = kinematics(m, s.qpos, body_pos)
  S = motion_subspace(m, kin)
  I = body_inertias(m, kin)
  M = mass_matrix(m, kin, S, I)
  bias = bias_forces(m, kin, s.qvel, S, I)
  passive = -m.jnt_damping * s.qvel
  act = np.zeros(m.nv)
  for a in range(len(m.act_joint)):
    j = m.act_joint[a]
    c = np.clip(s.ctrl[a], *m.act_ctrlrange[a])
    act[j] += m.act_kp[a] * (c - s.qpos[j])
  tau = passive + act - bias
  Minv = np.linalg.inv(M)
  a0 = Minv @ tau
  J, res, aref, R, is_eq = constraints(m, kin, S, s.qpos, s.qvel, s.mocap_pos, s.mocap_quat)
  f = np.zeros(0)
  qfrc_c = np.zeros(m.nv)
  if len(res):
    A = J @ Minv @ J.T
    f = solve_constraints(A, R, aref - J @ a0, is_eq)
    qfrc_c = J.T @ f
  qacc = a0 + Minv @ qfrc_c
  return dict(kin=kin, M=M, qacc=qacc, tau=tau, qfrc_constraint=qfrc_c, efc_force=f, efc_pos=res)


def step(m, s, body_pos=None):
  """one timestep: semi-implicit Euler, joint damping integrated implicitly"""
  out = forward(m, s, body_pos)
  M = out['M']
  Mh = M + m.dt * np.diag(m.jnt_damping)
  qacc = np.linalg.solve(Mh, M @ out['qacc'])
  s.qvel = s.qvel + m.dt * qacc
  s.qpos = s.qpos + m.dt * s.qvel
  return out


def dof_qpos0(m):
  """per-dof coordinates of the reference configuration: 0, except the translation of a free body (its qpos0 = body pos)"""
  q = np.zeros(m.nv)
  if hasattr(m, 'body_free'):
    for b in np.nonzero(m.body_free)[0]:
      q[m.body_dofs[b][:3]] = m.body_qpos0[b][:3]
  return q


def inverse_weights(m):
  """body_invweight0 [nb,2] (translational, rotational) and dof_invweight0 [nv] at qpos0 = 0 (MuJoCo compiles these
  into the model; the constraint regularizer uses them as the diagonal approximation of J M^-1 J^T)"""
  kin = kinematics(m, dof_qpos0(m))
  S = motion_subspace(m, kin)
  Minv = np.linalg.inv(mass_matrix(m, kin, S))
  bw = np.zeros((m.nb, 2))
  for b in range(1, m.nb):
    J = body_jacobian(m, S, b, kin['xipos'][b])
    A = J @ Minv @ J.T
    bw[b] = [np.trace(A[3:, 3:]) / 3, np.trace(A[:3, :3]) / 3]
  return bw, np.diag(Minv).copy()


# ======================================================================================================================
# Reduced ("link") model: one link per dof = the jointed body merged with its fixed descendants.  This is the form
# the HIP stepper consumes (earl_benchmark_amd/csrc/physics.hip); `LinkModel.step` is its line-by-line reference.
# ======================================================================================================================
def reduce_model(m, body_pos=None, attach_bodies=(), attach_sites=(), attach_geoms=(), collision=None, weld_translation_calibration=None):
  # (weld factors: m.weld_calibration, set by tools/mjcf_compile.py; weld_translation_calibration overrides the first one -- experiments)
  """-> dict of arrays: links in an order where parents precede children (the dof order of these models)"""
  bp = m.body_pos if body_pos is None else body_pos
  link_of_body = np.full(m.nb, -1)          # nearest moving ancestor-or-self link of each body (-1: world-fixed)
  for b in range(1, m.nb):
    if m.body_dofs[b]:
      link_of_body[b] = m.body_dofs[b][-1]  # a body with several dofs (free joint) is a chain of links; the LAST carries the body
    else:
      link_of_body[b] = link_of_body[m.body_parent[b]]
  # pose of every body in the frame of its link's jointed body (or world), with all joints at zero
  rel_pos, rel_quat = np.zeros((m.nb, 3)), np.tile([1.0, 0, 0, 0], (m.nb, 1))
  for b in range(1, m.nb):
    p = m.body_parent[b]
    if m.body_dofs[b]:
      continue                              # the jointed body itself is the link frame
    rel_pos[b] = rel_pos[p] + quat_mat(rel_quat[p]) @ bp[b]
    rel_quat[b] = quat_mul(rel_quat[p], m.body_quat[b])
  nv = m.nv
  out = dict(parent=np.full(nv, -1, np.int32), tpos=np.zeros((nv, 3)), tquat=np.tile([1.0, 0, 0, 0], (nv, 1)), jtype=m.jnt_type.astype(np.int32),
             jaxis=m.jnt_axis.copy(), jpos=m.jnt_pos.copy(), mass=np.zeros(nv), com=np.zeros((nv, 3)), inertia=np.zeros((nv, 6)))
  for l in range(nv):
    b = int(m.jnt_body[l]); p = int(m.body_parent[b])
    if l != m.body_dofs[b][0]:              # later dof of a multi-dof body: rides on the previous one, same frame, no mass
      out['parent'][l] = l - 1
    else:
      out['parent'][l] = link_of_body[p]
      out['tpos'][l] = rel_pos[p] + quat_mat(rel_quat[p]) @ bp[b]      # jointed body's frame in its parent link's frame
      out['tquat'][l] = quat_mul(rel_quat[p], m.body_quat[b])
    if l != m.body_dofs[b][-1]:
      continue
    members = [bb for bb in range(1, m.nb) if link_of_body[bb] == l]
    M = sum(m.body_mass[bb] for bb in members)
    c = sum(m.body_mass[bb] * (rel_pos[bb] + quat_mat(rel_quat[bb]) @ m.body_ipos[bb]) for bb in members) / M
    I = np.zeros((3, 3))
    for bb in members:
      R = quat_mat(rel_quat[bb]) @ quat_mat(m.body_iquat[bb])
      r = rel_pos[bb] + quat_mat(rel_quat[bb]) @ m.body_ipos[bb] - c
      I += R @ np.diag(m.body_inertia[bb]) @ R.T + m.body_mass[bb] * (r @ r * np.eye(3) - np.outer(r, r))
    out['mass'][l], out['com'][l] = M, c
    out['inertia'][l] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
  assert all(out['parent'][l] < l for l in range(nv)), 'links must be ordered parents first'
  att = []                                   # named points/frames rigidly attached to links
  for name in attach_bodies:
    b = m.body_id(name); att.append((name, link_of_body[b], rel_pos[b], rel_quat[b]))
  for name in attach_sites:
    s = m.site_id(name); b = int(m.site_body[s])
    att.append((name, link_of_body[b], rel_pos[b] + quat_mat(rel_quat[b]) @ m.site_pos[s], rel_quat[b]))
  for name in attach_geoms:
    g = m.geom_id(name); b = int(m.geom_body[g])
    att.append((name, link_of_body[b], rel_pos[b] + quat_mat(rel_quat[b]) @ m.geom_pos[g], rel_quat[b]))
  out['att_link'] = np.array([a[1] for a in att], np.int32)
  out['att_pos'] = np.stack([a[2] for a in att]); out['att_quat'] = np.stack([a[3] for a in att])
  out['att_names'] = np.array([a[0] for a in att])
  for k in ('jnt_limited', 'jnt_range', 'jnt_damping', 'jnt_armature', 'jnt_solref', 'jnt_solimp', 'dof_invweight0',
            'act_joint', 'act_kp', 'act_ctrlrange', 'weld_solref', 'weld_solimp', 'gravity', 'timestep'):
    out[k] = np.array(getattr(m, k))
  # optional joint tables (models compiled before round 2 do not carry them): dry friction, springs, actuator force limits, joint couplings
  out['jnt_frictionloss'] = np.array(getattr(m, 'jnt_frictionloss', np.zeros(nv)))
  out['jnt_stiffness'] = np.array(getattr(m, 'jnt_stiffness', np.zeros(nv)))
  out['jnt_springref'] = np.array(getattr(m, 'jnt_springref', np.zeros(nv)))
  out['act_forcerange'] = np.array(getattr(m, 'act_forcerange', np.tile([-np.inf, np.inf], (len(m.act_joint), 1)))).reshape(-1, 2)
  neq = len(getattr(m, 'jeq_joint1', ()))
  out['jeq_joint1'] = np.array(getattr(m, 'jeq_joint1', np.zeros(0)), np.int32)
  out['jeq_joint2'] = np.array(getattr(m, 'jeq_joint2', np.zeros(0)), np.int32)
  out['jeq_coef'] = np.array(getattr(m, 'jeq_coef', np.zeros((0, 2))), float).reshape(neq, 2)
  out['jeq_solref'] = np.array(getattr(m, 'jeq_solref', np.zeros((0, 2))), float).reshape(neq, 2)
  out['jeq_solimp'] = np.array(getattr(m, 'jeq_solimp', np.zeros((0, 5))), float).reshape(neq, 5)
  # mj_diagApprox, joint equality: dof_invweight0 of the two joints
  out['jeq_invweight'] = np.array([m.dof_invweight0[out['jeq_joint1'][e]] + m.dof_invweight0[out['jeq_joint2'][e]] for e in range(neq)])
  wb = int(m.weld_body2[0])
  out['weld_att'] = np.int32(list(out['att_names']).index(str(m.body_names[wb])))
  ftr = float(m.weld_calibration[0]) if weld_translation_calibration is None else float(weld_translation_calibration)
  out['weld_invweight'] = m.body_invweight0[wb] * np.array([ftr, float(m.weld_calibration[1])])
  out['weld_calibration'] = np.array(m.weld_calibration, float)
  out['cone_elliptic'] = np.int32(getattr(m, 'cone_elliptic', 0))        # <option cone=...>: the contact rows' friction cone (LinkModel.solve_primal)
  out['weld_mocap_quat'] = np.array(m.body_quat[int(m.weld_body1[0])])          # the mocap body's orientation in the model (its mocap_quat after sim.reset())
  out['weld_mocap_pos'] = np.array(m.body_pos[int(m.weld_body1[0])])
  # generalized coordinates: qpos has one entry per dof, except that a free body's orientation is a unit quaternion
  # stored where MuJoCo stores it (after the body's three translations): nq = nv + 1 per free body
  ball = [l for l in range(nv) if m.jnt_type[l] == 2]
  assert len(ball) <= 1 and (not ball or ball[0] == nv - 3), 'one free body, last in the dof order'
  out['ball_dof'] = np.int32(ball[0] if ball else -1)
  out['qpos0'] = np.concatenate([dof_qpos0(m)[:ball[0]], m.body_qpos0[m.jnt_body[ball[0]]][3:]]) if ball else dof_qpos0(m)
  if collision is not None:
    out.update(collision_primitives(m, link_of_body, rel_pos, rel_quat, collision, bp))
  return out


def collision_primitives(m, link_of_body, rel_pos, rel_quat, spec, body_pos_=None):
  """Collision geometry of the link model: SPHERES (cylinders become chains of spheres along their axis at spacing
  <= 0.75 r, hemispherical ends inside the cylinder; box corner POINTS are spheres of radius 0) tested against BOXES.
  spec: dict(plates=[geom names], chains=[geom ids], corner_sets=[[geom names merged into one box]], big_boxes=[geom ids]).
  Pairs: chain spheres x plates, corner points x big boxes.  Per-primitive solver parameters are the geoms' own."""
  sph, box = [], []

  def gframe(g):
    b = int(m.geom_body[g])
    return (int(link_of_body[b]), rel_pos[b] + quat_mat(rel_quat[b]) @ m.geom_pos[g], quat_mul(rel_quat[b], m.geom_quat[g]))

  def params(g):
    si = np.array(m.geom_solimp[g], float)
    if si[3] == 0 and si[4] == 0:                 # three-number solimp in the MJCF: midpoint / power keep their defaults
      si[3:] = [0.5, 2.0]
    return dict(mu=float(m.geom_friction[g][0]), solref=np.array(m.geom_solref[g], float), solimp=si,
                margin=float(m.geom_margin[g]), invw=float(m.body_invweight0[int(m.geom_body[g])][0]),
                condim=int(m.geom_condim[g]), mu_tor=float(m.geom_friction[g][1]))

  def union_box(names):
    """the (parallel) boxes `names` merged into one box in the first one's axes -> link, centre, quat, half sizes"""
    gs = [m.geom_id(n) for n in names]
    l, _, q = gframe(gs[0])
    Rq = quat_mat(q)
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for g in gs:
      l2, p2, q2 = gframe(g)
      assert l2 == l and np.allclose(q2, q)
      c = Rq.T @ p2
      lo, hi = np.minimum(lo, c - m.geom_size[g][:3]), np.maximum(hi, c + m.geom_size[g][:3])
    return l, Rq @ (0.5 * (lo + hi)), q, 0.5 * (hi - lo), lo, hi, Rq

  # gripper plates: a finger is a claw box and a pad box (same size, 3 mm apart) -> ONE box per finger with the pad's
  # contact parameters (the pad is the gripping face); otherwise every touching sphere would produce two contacts
  # Which spheres meet which boxes: every sphere belongs to a named SET, every box ACCEPTS a tuple of sets.  Defaults
  # (the door task): cylinder chains are set 'chain' and meet the plates; box corners are set 'corner' and meet the big boxes.
  for names in spec.get('plates', ()):
    if spec.get('plates_split'):
      # round 5: every geom of a finger is its own box with its own contact parameters -- the claw plate (class base_col) and the pad
      # (solimp 0.95 0.99 0.01, friction 2 0.1 0.002; metaworld_assets/objects/assets/xyz_base.xml:163,173,179,185), 3 mm apart, overlapping -- as MuJoCo sees them
      for nm in names:
        g = m.geom_id(nm)
        l, p, q = gframe(g)
        box.append(dict(link=l, pos=p, quat=q, half=np.array(m.geom_size[g][:3], float), accept=tuple(spec.get('plates_accept', ('chain',))), **params(g)))
      continue
    l, p, q, half, _, _, _ = union_box(names)
    box.append(dict(link=l, pos=p, quat=q, half=half, accept=tuple(spec.get('plates_accept', ('chain',))), **params(m.geom_id(names[-1]))))
  # boxes given explicitly (a finger built from a dozen capsules and a box is stood in for by ONE box): dict(body, pos, quat, half, like = geom
  # whose contact parameters it carries, accept); with 'corners': its eight corner points also become a sphere set of that name
  for e in [x for x in spec.get('explicit_boxes', ()) if not x.get('late')]:
    b = m.body_id(e['body'])
    l, p, q = int(link_of_body[b]), rel_pos[b] + quat_mat(rel_quat[b]) @ np.asarray(e['pos'], float), quat_mul(rel_quat[b], np.asarray(e.get('quat', [1.0, 0, 0, 0]), float))
    box.append(dict(link=l, pos=p, quat=q, half=np.asarray(e['half'], float), accept=tuple(e['accept']), **params(e['like'])))
    if e.get('corners'):
      for sx in (-1, 1):
        for sy in (-1, 1):
          for sz in (-1, 1):
            sph.append(dict(link=l, pos=p + quat_mat(q) @ (np.asarray(e['half'], float) * [sx, sy, sz]), r=0.0, set=e['corners'], **params(e['like'])))
  for e in spec.get('big_boxes', ()):
    g, accept = (e['geom'], tuple(e['accept'])) if isinstance(e, dict) else (e, ('corner',))
    l, p, q = gframe(g)
    box.append(dict(link=l, pos=p, quat=q, half=np.array(m.geom_size[g][:3], float), accept=accept, **params(g)))
  for e in spec.get('chains', ()):
    g, sname, spacing = (e['geom'], e.get('set', 'chain'), e.get('spacing', 0.75)) if isinstance(e, dict) else (e, 'chain', 0.75)
    l, p, q = gframe(g)
    if m.geom_type[g] == 4:                        # a long box stood in for by the chain of its inscribed spheres along its longest axis
      sz = np.array(m.geom_size[g][:3], float)
      ax = int(np.argmax(sz))
      r, h = float(np.min(np.delete(sz, ax))), float(sz[ax])
      d = np.eye(3)[ax]
    else:                                          # cylinder / capsule: axis z, size = (radius, half length)
      r, h, d = float(m.geom_size[g][0]), float(m.geom_size[g][1]), np.array([0.0, 0, 1])
      if m.geom_type[g] == 2:
        h += r                                     # a capsule's hemispherical caps lie beyond its half length
    span = max(h - r, 0.0)
    k = 1 if span == 0 else int(np.ceil(2 * span / (spacing * r))) + 1
    for t in (np.linspace(-span, span, k) if k > 1 else [0.0]):
      sph.append(dict(link=l, pos=p + quat_mat(q) @ (d * t), r=r, set=sname, chain=(g, p.copy(), quat_mat(q) @ d, span), **params(g)))
  # round 5: CYLINDERS as cylinders -- flat ends, one contact per (box, cylinder) pair from the box-cylinder narrow phase (MuJoCo sends
  # this pair to its general convex routine, which returns one contact).  Stored in the sphere list: pos = centre, dir = axis, hl = half length, r = radius
  # (an edge has r = 0, a sphere hl = 0); pairs of kind 2 (col_pair_kind).
  for e in spec.get('cylinders', ()):
    g, sname = (e['geom'], e.get('set', 'cyl')) if isinstance(e, dict) else (e, 'cyl')
    l, p, q = gframe(g)
    assert m.geom_type[g] == 3
    sph.append(dict(link=l, pos=p, r=float(m.geom_size[g][0]), dir=quat_mat(q)[:, 2].copy(), hl=float(m.geom_size[g][1]), set=sname, cyl=True, **params(g)))
  if spec.get('drop_contained'):
    # a chain sphere that lies wholly inside ANOTHER chain's capsule on the same link (the door handle's thin rod runs through a fatter
    # sleeve) can only touch what the outer capsule already touches: dropped, which keeps the plate blocks at one 16-lane pass
    def contained(a):
      for b2 in sph:
        if 'chain' in b2 and b2['chain'][0] != a['chain'][0] and b2['link'] == a['link'] and b2['r'] >= a['r']:
          _, c0, ax, sp = b2['chain']
          t = float(np.clip((a['pos'] - c0) @ ax, -sp, sp))
          if np.linalg.norm(a['pos'] - (c0 + t * ax)) + a['r'] <= b2['r'] + 1e-12:
            return True
      return False
    sph[:] = [a for a in sph if 'chain' not in a or not contained(a)]
  # EDGES vs CAPSULES.  A chain sphere touching the flat of a plate is pushed along the plate's normal, whatever the angle between
  # the plate and the cylinder the chain stands for; MuJoCo's box-cylinder contact sits where the plate's EDGE presses into the
  # cylinder's side and pushes along the cylinder's radial direction.  So the long edges of the plates (segments; stored as
  # "spheres" of radius 0 with a direction and a half length) are tested against the cylinders as capsules (stored as "boxes" of
  # kind 1: axis = local z, half = (r, r, h), hemispherical ends inside the cylinder like the chains').  Closest points of two
  # segments; normal from the capsule's axis to the edge.  spec['edge_caps'] = dict(plates=[[geom names]], caps=[geom ids], set=name)
  ec = spec.get('edge_caps')
  if ec:
    for g in ec['caps']:
      l, p, q = gframe(g)
      assert m.geom_type[g] == 3
      r, h = float(m.geom_size[g][0]), float(m.geom_size[g][1])
      box.append(dict(link=l, pos=p, quat=q, half=np.array([r, r, h]), kind=1,
                      accept=tuple(ec.get('set', 'edge') + str(k) for k in range(len(ec['plates']))) if ec.get('per_plate') else (ec.get('set', 'edge'),), **params(g)))
    for k, names in enumerate(ec['plates']):
      l, p, q, half, _, _, Rq = union_box(names)
      ax = int(np.argmax(half))                     # the long edges run along the plate's longest axis
      sname = ec.get('set', 'edge') + (str(k) if ec.get('per_plate') else '')   # per_plate: one set (one bounding test) per plate
      o1, o2 = [a for a in range(3) if a != ax]
      for s1 in (-1, 1):
        for s2 in (-1, 1):
          off = np.zeros(3); off[o1], off[o2] = s1 * half[o1], s2 * half[o2]
          sph.append(dict(link=l, pos=p + Rq @ off, r=0.0, dir=Rq[:, ax].copy(), hl=float(half[ax]), set=sname, **params(m.geom_id(names[-1]))))
  for e in spec.get('corner_sets', ()):
    names, sname = (e['names'], e.get('set', 'corner')) if isinstance(e, dict) else (e, 'corner')
    l, _, q, _, lo, hi, Rq = union_box(names)
    gs = [m.geom_id(names[0])]
    for sx in (0, 1):
      for sy in (0, 1):
        for sz in (0, 1):
          c = np.array([hi[0] if sx else lo[0], hi[1] if sy else lo[1], hi[2] if sz else lo[2]])
          sph.append(dict(link=l, pos=Rq @ c, r=0.0, set=sname, **params(gs[0])))
  # additions of later rounds come LAST in both lists, so that the pair and block order of everything before them -- the priority order of the
  # contact slots -- stays what it was: explicit boxes flagged `late`, and spheres given explicitly (a mesh hull stood in for by a few spheres:
  # dict(body, pos, r, set, like = geom whose contact parameters they carry))
  for e in [x for x in spec.get('explicit_boxes', ()) if x.get('late')]:
    b = m.body_id(e['body'])
    box.append(dict(link=int(link_of_body[b]), pos=rel_pos[b] + quat_mat(rel_quat[b]) @ np.asarray(e['pos'], float),
                    quat=quat_mul(rel_quat[b], np.asarray(e.get('quat', [1.0, 0, 0, 0]), float)), half=np.asarray(e['half'], float), accept=tuple(e['accept']), **params(e['like'])))
  for e in spec.get('explicit_spheres', ()):
    b = m.body_id(e['body'])
    sph.append(dict(link=int(link_of_body[b]), pos=rel_pos[b] + quat_mat(rel_quat[b]) @ np.asarray(e['pos'], float), r=float(e['r']), set=e['set'], **params(e['like'])))
  # Permanent deep box-in-box contacts (the door panel stands 2.3 cm inside the table top: obj_init_pos z = 0.1, panel
  # half height 0.123).  Such a contact has all four pyramid edges active and a constant depth, and the moving box has
  # one dof, so its 4 x 4 edges reduce EXACTLY to one soft velocity row on that dof (tests/test_physics.py checks the
  # reduction against the explicit contacts):  cost 1/2 G (a_j + b v_j)^2,  G = sum_k rho_k^2 / R0.
  nv = len(m.jnt_body)
  drag_G, drag_b = np.zeros(nv), np.zeros(nv)
  drag_calib = float(spec.get('drag_calibration', 1.0))   # tools/mjcf_compile.py DOOR_DRAG_CALIBRATION
  for g_mov, g_fix, depth in spec.get('drag', ()):
    l, p, q = gframe(g_mov)
    a, b = params(g_mov), params(g_fix)
    solref, solimp, margin = 0.5 * (a['solref'] + b['solref']), 0.5 * (a['solimp'] + b['solimp']), max(a['margin'], b['margin'])
    kk, bb, dd = kbimp(solref, solimp, -depth - margin, float(m.timestep))
    R0 = max((1 - dd) / dd * (a['invw'] + b['invw']), 1e-15)
    h = np.array(m.geom_size[g_mov][:3], float)
    axis = m.jnt_axis[l]; anchor = m.jnt_pos[l]
    assert m.jnt_type[l] == 0 and abs(abs(quat_mat(q)[:, 2] @ axis) - 1) < 1e-9, 'drag reduction: hinge parallel to the box z axis'
    for sx in (-1, 1):
      for sy in (-1, 1):
        c = p + quat_mat(q) @ np.array([sx * h[0], sy * h[1], -h[2]]) - anchor
        rho2 = c @ c - (c @ axis) ** 2
        drag_G[l] += drag_calib * rho2 / R0
    drag_b[l] = bb
  # pairs in BOX-MAJOR order, cut into blocks (one box x one set of spheres) that carry a bounding test: the kernel skips
  # a whole block when the set's bounding sphere is clear of the box's.  Chain spheres ride on one link; the corner
  # points of both gripper plates are bounded together in the frame of their common parent (the claws slide: slack).
  pairs, blocks = [], []
  parent = [int(link_of_body[m.body_parent[int(m.jnt_body[l])]]) for l in range(nv)]
  for j, bx in enumerate(box):
   for sname in bx['accept']:
    members = [i for i, sp in enumerate(sph) if sp['set'] == sname and sp['link'] != bx['link']]
    if not members:
      continue
    links = sorted(set(sph[i]['link'] for i in members))
    if len(links) == 1:
      bl, slack = links[0], 0.0
      pts = np.array([sph[i]['pos'] for i in members])
      dirs = np.array([sph[i].get('dir', np.zeros(3)) for i in members])
    else:                                         # sets on sibling slide links: bound them in the parent's frame at mid range
      bl = parent[links[0]]
      assert all(parent[l] == bl and m.jnt_type[l] == 1 for l in links)
      slack = max(0.5 * (m.jnt_range[l][1] - m.jnt_range[l][0]) for l in links)
      pts, dirs = [], []
      for i in members:
        l = sph[i]['link']
        jb = int(m.jnt_body[l]); pb = int(m.body_parent[jb])
        tp = rel_pos[pb] + quat_mat(rel_quat[pb]) @ (m.body_pos if body_pos_ is None else body_pos_)[jb]
        tq = quat_mul(rel_quat[pb], m.body_quat[jb])
        mid = 0.5 * (m.jnt_range[l][0] + m.jnt_range[l][1])
        pts.append(tp + quat_mat(tq) @ (sph[i]['pos'] + m.jnt_axis[l] * mid))
        dirs.append(quat_mat(tq) @ sph[i].get('dir', np.zeros(3)))
      pts, dirs = np.array(pts), np.array(dirs)
    ctr = pts.mean(0)
    rad = max(np.sqrt(((pts[k] - ctr) ** 2).sum()) + sph[i]['r'] + sph[i].get('hl', 0.0) for k, i in enumerate(members)) + slack
    mmax = max(max(sph[i]['margin'], bx['margin']) for i in members)
    # second bound: the set's box in the frame of `bl` (axis-aligned there; radii, edge half-lengths and the slide slack included, the
    # margin too), tested against the block's box by the six face axes of the two boxes (a separating axis = no pair within the margin)
    ext = np.array([sph[i]['r'] + np.abs(dirs[k]) * sph[i].get('hl', 0.0) for k, i in enumerate(members)])
    lo, hi = (pts - ext).min(0) - slack, (pts + ext).max(0) + slack
    blocks.append(dict(box=j, link=bl, center=ctr, reach=rad + mmax + 1e-6, set=sname, members=members,
                       obb_center=0.5 * (lo + hi), obb_half=0.5 * (hi - lo) + mmax + 1e-6))
  # Contacts are kept first come, first served up to max_contacts, so the ORDER of the blocks is a priority: blocks of the sets
  # listed first in spec['set_priority'] come first (stable otherwise), and a block contributes at most spec['set_cap'][set]
  # contacts.  Without this a gripper standing in the hole block (16 plate corners in contact) used up every slot and the peg,
  # untouched on the table, fell through it.
  prio = list(spec.get('set_priority', ()))
  blocks.sort(key=lambda b: prio.index(b['set']) if b['set'] in prio else len(prio))
  for b in blocks:
    b['begin'], b['end'] = len(pairs), len(pairs) + len(b['members'])
    b['cap'] = int(spec.get('set_cap', {}).get(b['set'], spec.get('max_contacts', 8)))
    pairs += [(i, b['box']) for i in b['members']]
  cls, pair_cls = [], []
  for i, j in pairs:                             # contact parameters: MuJoCo mixes the two geoms (max friction / margin, mean solref / solimp)
    a, b = sph[i], box[j]
    key = (max(a['mu'], b['mu']), tuple(0.5 * (a['solref'] + b['solref'])), tuple(0.5 * (a['solimp'] + b['solimp'])),
           max(a['margin'], b['margin']), a['invw'] + b['invw'])
    if spec.get('torsion'):                      # condim 4 pairs (MuJoCo: max of the two geoms' condim, elementwise max of their friction): torsional coefficient; 0 = condim 3
      key = key + ((max(a['mu_tor'], b['mu_tor']) if max(a['condim'], b['condim']) >= 4 else 0.0),)
    if key not in cls:
      cls.append(key)
    pair_cls.append(cls.index(key))
  return dict(col_sph_link=np.array([x['link'] for x in sph], np.int32), col_sph_pos=np.array([x['pos'] for x in sph]),
              col_sph_r=np.array([x['r'] for x in sph]),
              col_sph_dir=np.array([x.get('dir', np.zeros(3)) for x in sph]), col_sph_hl=np.array([x.get('hl', 0.0) for x in sph]),
              col_box_kind=np.array([x.get('kind', 0) for x in box], np.int32),
              col_box_link=np.array([x['link'] for x in box], np.int32), col_box_pos=np.array([x['pos'] for x in box]),
              col_box_quat=np.array([x['quat'] for x in box]), col_box_half=np.array([x['half'] for x in box]),
              col_pair=np.array(pairs, np.int32).reshape(-1, 2), col_pair_cls=np.array(pair_cls, np.int32),
              col_cls_mu=np.array([c[0] for c in cls]), col_cls_solref=np.array([c[1] for c in cls]),
              col_cls_solimp=np.array([c[2] for c in cls]), col_cls_margin=np.array([c[3] for c in cls]),
              col_cls_invw=np.array([c[4] for c in cls]), col_cls_mu_tor=np.array([(c[5] if len(c) > 5 else 0.0) for c in cls]),
              col_pair_kind=np.array([(2 if sph[i].get('cyl') else (1 if box[j].get('kind', 0) == 1 else 0)) for i, j in pairs], np.int32), dof_drag_G=drag_G, dof_drag_b=drag_b, dof_drag_calibration=np.float64(drag_calib),
              col_blk_begin=np.array([b['begin'] for b in blocks], np.int32), col_blk_end=np.array([b['end'] for b in blocks], np.int32),
              col_blk_box=np.array([b['box'] for b in blocks], np.int32), col_blk_link=np.array([b['link'] for b in blocks], np.int32),
              col_blk_center=np.array([b['center'] for b in blocks]), col_blk_reach=np.array([b['reach'] for b in blocks]),
              col_blk_cap=np.array([b['cap'] for b in blocks], np.int32),
              col_blk_obb_center=np.array([b['obb_center'] for b in blocks]), col_blk_obb_half=np.array([b['obb_half'] for b in blocks]),
              max_contacts=np.int32(spec.get('max_contacts', 8)))


def sym6(v):
  return np.array([[v[0], v[3], v[4]], [v[3], v[1], v[5]], [v[4], v[5], v[2]]])


class LinkModel:
  """the reduced tables + the reference of the HIP stepper (same phases, same order of operations where it matters)"""

  def __init__(self, path_or_dict):
    d = np.load(path_or_dict, allow_pickle=False) if isinstance(path_or_dict, str) else path_or_dict
    for k in (d.files if hasattr(d, 'files') else d):
      setattr(self, k, np.array(d[k]))
    self.nv = len(self.parent)
    self.dt = float(self.timestep)
    self.max_contacts = int(self.max_contacts)
    self.elliptic = bool(int(getattr(self, 'cone_elliptic', 0)))       # table 'cone_elliptic': the MJCF's <option cone="elliptic"> (door, peg: metaworld's basic_scene.xml:2)
    # qpos index of dof l: with a free body the orientation quaternion takes FOUR qpos slots at ball_dof .. ball_dof + 3 (MuJoCo's layout), so the
    # dofs behind its three rotation dofs sit one slot further (a free ROOT body -- the minitaur's base: qpos = [xyz, quat, joints]); a free body whose
    # rotation dofs are the last three (the peg) has none behind it
    bd_ = int(getattr(self, 'ball_dof', -1))
    self.qadr = np.array([l + 1 if (bd_ >= 0 and l > bd_ + 2) else l for l in range(self.nv)])
    self.anc = []
    for l in range(self.nv):
      a, p = [l], self.parent[l]
      while p >= 0:
        a = [int(p)] + a
        p = self.parent[p]
      self.anc.append(a)

  def kinematics(self, qpos):
    """joint types: 0 hinge, 1 slide, 2 / 3 the rotation of a free body (MuJoCo: angular velocity in BODY axes, orientation a
    unit quaternion in qpos): link type 2 applies the quaternion and its axis is the rotated body x axis, the two type-3
    links that follow do not move and contribute the body y and z axes."""
    nv = self.nv
    pos, quat = np.zeros((nv, 3)), np.zeros((nv, 4))
    S = np.zeros((nv, 6))
    for l in range(nv):
      p = self.parent[l]
      pp, pq = (pos[p], quat[p]) if p >= 0 else (np.zeros(3), np.array([1.0, 0, 0, 0]))
      x = pp + quat_mat(pq) @ self.tpos[l]
      q = quat_mul(pq, self.tquat[l])
      R = quat_mat(q)
      anchor, axis = x + R @ self.jpos[l], R @ self.jaxis[l]
      if self.jtype[l] == 0:
        h = 0.5 * qpos[self.qadr[l]]
        q = quat_mul(q, np.concatenate([[np.cos(h)], np.sin(h) * self.jaxis[l]]))
        x = anchor - quat_mat(q) @ self.jpos[l]
        S[l, :3], S[l, 3:] = axis, np.cross(anchor, axis)
      elif self.jtype[l] == 1:
        x = x + axis * qpos[self.qadr[l]]
        S[l, 3:] = axis
      else:
        if self.jtype[l] == 2:
          qb = np.asarray(qpos[l:l + 4], float)
          q = quat_mul(q, qb / np.sqrt(qb @ qb))         # normalised, as mj_kinematics does
          x = anchor - quat_mat(q) @ self.jpos[l]
          axis = quat_mat(q) @ self.jaxis[l]
        S[l, :3], S[l, 3:] = axis, np.cross(anchor, axis)
      pos[l], quat[l] = x, q
    return pos, quat, S

  def integrate_pos(self, qpos, qvel):
    """qpos <- qpos (+) dt qvel: per-dof addition; a free body's quaternion by mju_quatIntegrate (q * exp(dt w / 2), w in body axes)"""
    bd = int(getattr(self, 'ball_dof', -1))
    if bd < 0:
      return qpos + self.dt * qvel
    out = np.array(qpos, float)
    for l in range(self.nv):
      if not bd <= l <= bd + 2:
        out[self.qadr[l]] = qpos[self.qadr[l]] + self.dt * qvel[l]
    w = qvel[bd:bd + 3]
    nw = np.sqrt(w @ w)
    qn = qpos[bd:bd + 4] / np.sqrt(qpos[bd:bd + 4] @ qpos[bd:bd + 4])
    if nw > 0:
      ang = self.dt * nw
      qn = quat_mul(qn, np.concatenate([[np.cos(0.5 * ang)], np.sin(0.5 * ang) * w / nw]))
    out[bd:bd + 4] = qn / np.sqrt(qn @ qn)
    return out

  def attachment(self, pos, quat, k):
    l = self.att_link[k]
    if l < 0:
      return self.att_pos[k].copy(), self.att_quat[k].copy()
    return pos[l] + quat_mat(quat[l]) @ self.att_pos[k], quat_mul(quat[l], self.att_quat[k])

  def forward(self, qpos, qvel, ctrl, mocap_pos, mocap_quat, a_prev=None, qfrc=None):
    """qfrc: generalized forces applied from outside (the minitaur's motor torques, computed by the env's motor model per timestep), or None"""
    nv = self.nv
    pos, quat, S = self.kinematics(qpos)
    I6 = []
    for l in range(nv):
      R = quat_mat(quat[l])
      I6.append(spatial_inertia(self.mass[l], pos[l] + R @ self.com[l], R @ sym6(self.inertia[l]) @ R.T))
    Ic = [x.copy() for x in I6]
    for l in range(nv - 1, -1, -1):
      if self.parent[l] >= 0:
        Ic[self.parent[l]] += Ic[l]
    M = np.zeros((nv, nv))
    for i in range(nv):
      F = Ic[i] @ S[i]
      for j in self.anc[i]:
        M[i, j] = M[j, i] = S[j] @ F
    M[np.arange(nv), np.arange(nv)] += self.jnt_armature
    V, A, F = np.zeros((nv, 6)), np.zeros((nv, 6)), np.zeros((nv, 6))
    a_base = np.concatenate([np.zeros(3), -self.gravity])
    for l in range(nv):
      p = self.parent[l]
      Vp, Ap = (V[p], A[p]) if p >= 0 else (np.zeros(6), a_base)
      # d/dt of the axis: V_parent x S_l; the three rotation axes of a free body all use the velocity BEFORE any of the
      # three (mj_comVel: "compute all 3 dofdots before updating the velocity")
      pc = p if self.jtype[l] != 3 else self.parent[l - 1 if self.jtype[l - 1] == 2 else l - 2]
      Vc = V[pc] if pc >= 0 else np.zeros(6)
      A[l] = Ap + crossm(Vc) @ S[l] * qvel[l]
      V[l] = Vp + S[l] * qvel[l]
      F[l] = I6[l] @ A[l] + crossf(V[l]) @ (I6[l] @ V[l])
    for l in range(nv - 1, -1, -1):
      if self.parent[l] >= 0:
        F[self.parent[l]] += F[l]
    bias = np.array([S[l] @ F[l] for l in range(nv)])
    tau = -self.jnt_damping * qvel - bias
    if hasattr(self, 'jnt_stiffness'):             # joint springs (mj_passive): -k (q - springref)
      tau = tau - self.jnt_stiffness * (np.asarray(qpos[:nv], float) - self.jnt_springref)
    for a in range(len(self.act_joint)):
      j = self.act_joint[a]
      frc = self.act_kp[a] * (np.clip(ctrl[a], *self.act_ctrlrange[a]) - qpos[j])
      if hasattr(self, 'act_forcerange'):          # forcelimited actuators (mj_fwdActuation clamps the actuator force)
        frc = float(np.clip(frc, *self.act_forcerange[a]))
      tau[j] += frc
    if qfrc is not None:
      tau = tau + np.asarray(qfrc, float)
    a0 = np.linalg.solve(M, tau)
    # constraint rows: 6 weld rows (equalities), then lower / upper limit of every dof, then 4 pyramid edges per contact
    k = int(self.weld_att)
    has_weld = k >= 0                                        # (the minitaur model has no mocap weld: its six rows are never instantiated)
    hp, hq = self.attachment(pos, quat, k) if has_weld else (np.zeros(3), np.array([1.0, 0, 0, 0]))
    nlim = 6 + 2 * nv
    J = np.zeros((nlim, nv)); res = np.zeros(nlim); inst = np.zeros(nlim, bool)
    solref = [self.weld_solref[0]] * 6; solimp = [self.weld_solimp[0]] * 6
    invw = [self.weld_invweight[0]] * 3 + [self.weld_invweight[1]] * 3
    # weld rows as mj_instantiateEqual builds them (body1 = mocap, body2 = hand, relpose = identity after metaworld's
    # reset_mocap_welds): position error mocap - hand; orientation error = vector part of e = conj(q_hand) * q_mocap with
    # the EXACT Jacobian of that vector part, 0.5 * vec(conj(q_hand) * [0, -w_j] * q_mocap) = -0.5 * (e_w a + a x e_v),
    # a = R_hand^T w_j.  No sign flip for e_w < 0 (MuJoCo has none).
    q1 = np.asarray(mocap_quat, dtype=np.float64)           # as given, not normalised (rule replacing round 1's rotational factor)
    e = quat_mul(quat_conj(hq), q1)
    Rh = quat_mat(hq)
    for j in (self.anc[self.att_link[k]] if has_weld else ()):
      J[0:3, j] = -(S[j, 3:] + np.cross(S[j, :3], hp))
      a = Rh.T @ S[j, :3]
      J[3:6, j] = -0.5 * (e[0] * a + np.cross(a, e[1:]))
    res[0:3] = (mocap_pos - hp) if has_weld else 0.0
    res[3:6] = e[1:] if has_weld else 0.0
    inst[:6] = has_weld
    for j in range(nv):
      J[6 + 2 * j, j], J[7 + 2 * j, j] = 1.0, -1.0
      res[6 + 2 * j], res[7 + 2 * j] = qpos[self.qadr[j]] - self.jnt_range[j, 0], self.jnt_range[j, 1] - qpos[self.qadr[j]]
      inst[6 + 2 * j] = bool(self.jnt_limited[j]) and res[6 + 2 * j] < 0
      inst[7 + 2 * j] = bool(self.jnt_limited[j]) and res[7 + 2 * j] < 0
      solref += [self.jnt_solref[j]] * 2; solimp += [self.jnt_solimp[j]] * 2; invw += [self.dof_invweight0[j]] * 2
    aref, Rg = np.zeros(nlim), np.zeros(nlim)
    Jv = J @ qvel
    for i in range(nlim):
      kk, bb, dd = kbimp(solref[i], solimp[i], res[i], self.dt)
      aref[i] = -bb * Jv[i] - kk * dd * res[i]
      Rg[i] = max((1 - dd) / dd * invw[i], 1e-15)
    if self.contacts and hasattr(self, 'dof_drag_G'):
      for j in np.nonzero(self.dof_drag_G)[0]:   # soft velocity row of a permanent dragging contact: 1/2 G (a_j + b v_j)^2
        row = np.zeros(nv); row[j] = 1.0
        J = np.vstack([J, row]); aref = np.append(aref, -self.dof_drag_b[j] * qvel[j]); Rg = np.append(Rg, 1.0 / self.dof_drag_G[j])
        inst = np.append(inst, True); res = np.append(res, 0.0)
    # joint couplings (mjEQ_JOINT, linear: q1 - c0 - c1 q2 = 0 with qpos0 = 0): soft equality rows, regulariser from both dofs' invweight
    for e in range(len(getattr(self, 'jeq_joint1', ()))):
      j1, j2 = int(self.jeq_joint1[e]), int(self.jeq_joint2[e])
      c0, c1 = self.jeq_coef[e]
      row = np.zeros(nv); row[j1], row[j2] = 1.0, -c1
      r_ = qpos[j1] - c0 - c1 * qpos[j2]
      kk, bb, dd = kbimp(self.jeq_solref[e], self.jeq_solimp[e], r_, self.dt)
      J = np.vstack([J, row]); aref = np.append(aref, -bb * (row @ qvel) - kk * dd * r_)
      Rg = np.append(Rg, max((1 - dd) / dd * self.jeq_invweight[e], 1e-15)); inst = np.append(inst, True); res = np.append(res, r_)
    # connect constraints (MuJoCo mjEQ_CONNECT; Bullet's JOINT_POINT2POINT of the minitaur's four knee closures, envs/minitaur.py:212-217): the world
    # positions of two attachments coincide -- three soft equality rows, residual p1 - p2, Jacobian Jp(link1, p1) - Jp(link2, p2)
    for e in range(len(getattr(self, 'con_att1', ()))):
      k1, k2 = int(self.con_att1[e]), int(self.con_att2[e])
      p1, _ = self.attachment(pos, quat, k1)
      p2, _ = self.attachment(pos, quat, k2)
      Jp = np.zeros((3, nv))
      for kk_, pp_, sg in ((k1, p1, 1.0), (k2, p2, -1.0)):
        for j in self.anc[self.att_link[kk_]]:
          Jp[:, j] += sg * (S[j, 3:] + np.cross(S[j, :3], pp_))
      r3 = p1 - p2
      for c in range(3):
        kk, bb, dd = kbimp(self.con_solref[e], self.con_solimp[e], r3[c], self.dt)
        J = np.vstack([J, Jp[c]]); aref = np.append(aref, -bb * (Jp[c] @ qvel) - kk * dd * r3[c])
        Rg = np.append(Rg, max((1 - dd) / dd * self.con_invweight[e], 1e-15)); inst = np.append(inst, True); res = np.append(res, r3[c])
    n_eq_extra = len(aref) - nlim
    # dry joint friction (mjCNSTR_FRICTION_DOF): one row per dof with frictionloss, residual 0, force bounded by +- frictionloss
    fric = []
    for j in np.nonzero(getattr(self, 'jnt_frictionloss', np.zeros(nv)))[0]:
      kk, bb, dd = kbimp(self.jnt_solref[j], self.jnt_solimp[j], 0.0, self.dt)
      fric.append((int(j), -bb * qvel[j], max((1 - dd) / dd * self.dof_invweight0[j], 1e-15), float(self.jnt_frictionloss[j])))
    contacts = self.collide(pos, quat) if (self.contacts and hasattr(self, 'col_pair')) else []
    cone_mu = None
    if contacts:
      Jc, arc, Rc = self.contact_rows(contacts, S, qvel)
      J, aref, Rg = np.vstack([J, Jc]), np.concatenate([aref, arc]), np.concatenate([Rg, Rc])
      inst = np.concatenate([inst, np.ones(len(arc), bool)])
      if self.elliptic:
        cone_mu = [float(getattr(self, 'link_mu', {}).get(c['ls'], self.col_cls_mu[c['cls']])) for c in contacts]     # the last 3 rows per contact are (n, t1, t2)
    is_eq = np.zeros(len(aref), bool); is_eq[:6] = has_weld; is_eq[nlim:nlim + n_eq_extra] = True
    kw = {} if a_prev is None else dict(a_prev=a_prev)
    if cone_mu is not None:
      kw['cone_mu'] = cone_mu
    qacc, act = self.solve_primal(M, tau, J[inst], aref[inst], 1.0 / Rg[inst], is_eq[inst], fric=fric, **kw)
    f = np.zeros(len(aref)); active = np.zeros(len(aref), bool)
    idx = np.nonzero(inst)[0]
    f[idx] = np.where(act, -(J[idx] @ qacc - aref[idx]) / Rg[idx], 0.0)
    active[idx] = act
    return dict(pos=pos, quat=quat, M=M, qacc=qacc, f=f, res=res, active=active, contacts=contacts, a0=a0)

  block_cull = True        # False: test every pair (tests check that the cull never drops a contact)
  contacts = True          # class-level switch: LinkModel.contacts = False gives the contact-free stepper
  max_contacts = 8         # the kernel's cap (tables: 'max_contacts' = earl_collision_model.max_con): the first max_contacts active pairs in pair order

  def solve_primal(self, M, tau, J, aref, D, is_eq, iters=8, fric=(), a_prev=None, cone_mu=None):
    """MuJoCo's primal problem  min_a 1/2 (a-a0)' M (a-a0) + sum_r 1/2 D_r [J_r a - aref_r]_-^2  (equalities: both signs)
    by the active-set Newton iteration the kernel runs: rows start active, then active <=> J_r a < aref_r.
    fric: dry-friction rows (dof j, aref, R, loss): cost 1/2 x^2 / R for |x| <= R loss, loss (|x| - R loss / 2) beyond (x = a_j - aref): a
    row is QUADRATIC (state 0: it adds 1 / R to the diagonal) or saturated (state +-1: it pushes with -+loss); rows start quadratic.
    a_prev (warm start; the kernels pass the previous timestep's solution within one env step): the iteration starts from the set the rows
    take AT a_prev instead.  The fixed point is the same, so is the result; only the number of iterations changes.
    cone_mu (round 4, models compiled from an MJCF with cone="elliptic"): the last 3 len(cone_mu) rows are contacts as (normal, t1, t2) triples with ONE
    regulariser each (impratio 1) and MuJoCo's elliptic cost: with r = J a - aref, rho = |r_t|: nothing for r_n >= mu rho (top zone: separating), 1/2 D |r|^2 for
    rho <= -mu r_n (bottom zone: sticking), 1/2 D / (1 + mu^2) (r_n - mu rho)^2 in between (middle zone: sliding, force on the cone's surface).  A contact carries its zone and
    the row values J a of the iterate the zone was read from; the middle zone is linearised there (exact Hessian of the cost, Newton without line search); the iteration
    stops when no row, no friction row and no zone changes and every sliding contact's row values moved by less than ELL_TOL (relative) -- or after `iters` passes.
    Cold start: every contact in the bottom zone."""
    if cone_mu is not None:
      return self.solve_primal_elliptic(M, tau, J, aref, D, is_eq, iters, fric, a_prev, cone_mu)
    act = np.ones(len(aref), bool)
    fs = np.zeros(len(fric), int)
    if a_prev is not None:
      act = is_eq | (J @ a_prev - aref < 0)       # (the dry-friction rows keep their cold start: from a_prev's zones the iteration cycles far more often)
    a = None
    for _ in range(iters):
      Ja = J[act]
      H = M + Ja.T @ (D[act, None] * Ja)
      g = tau + Ja.T @ (D[act] * aref[act])
      for k, (j, ar, R, loss) in enumerate(fric):
        if fs[k] == 0:
          H[j, j] += 1.0 / R
          g[j] += ar / R
        else:
          g[j] -= fs[k] * loss
      a = np.linalg.solve(H, g)
      want = is_eq | (J @ a - aref < 0)
      nfs = np.array([0 if abs(a[j] - ar) <= R * loss else (1 if a[j] - ar > 0 else -1) for (j, ar, R, loss) in fric], int)
      if (want == act).all() and (nfs == fs).all():
        break
      act, fs = want, nfs
    return a, act

  ELL_TOL = 1e-8

  @staticmethod
  def cone_zone(r, mu):
    rho = np.sqrt(r[1] * r[1] + r[2] * r[2])
    return 0 if r[0] >= mu * rho else (1 if rho <= -mu * r[0] else 2)

  def solve_primal_elliptic(self, M, tau, J, aref, D, is_eq, iters, fric, a_prev, cone_mu):
    nc = len(cone_mu); nr = len(aref) - 3 * nc
    Ju, aru, Du, eq = J[:nr], aref[:nr], D[:nr], is_eq[:nr]
    warm = a_prev is not None
    act = (eq | (Ju @ a_prev - aru < 0)) if warm else np.ones(nr, bool)
    fs = np.zeros(len(fric), int)
    zone = np.ones(nc, int); Jak = np.zeros((nc, 3))
    if warm:
      for c in range(nc):
        Jak[c] = J[nr + 3 * c: nr + 3 * c + 3] @ a_prev
        zone[c] = self.cone_zone(Jak[c] - aref[nr + 3 * c: nr + 3 * c + 3], cone_mu[c])
    a = None
    for _ in range(iters):
      Ja = Ju[act]
      H = M + Ja.T @ (Du[act, None] * Ja)
      g = tau + Ja.T @ (Du[act] * aru[act])
      for k, (j, ar, R, loss) in enumerate(fric):
        if fs[k] == 0:
          H[j, j] += 1.0 / R
          g[j] += ar / R
        else:
          g[j] -= fs[k] * loss
      for c in range(nc):
        if zone[c] == 0:
          continue
        Jc, ac, Dn, mu = J[nr + 3 * c: nr + 3 * c + 3], aref[nr + 3 * c: nr + 3 * c + 3], D[nr + 3 * c], cone_mu[c]
        if zone[c] == 1:
          Hc, h = Dn * np.eye(3), Dn * ac
        else:
          r = Jak[c] - ac
          rho = np.sqrt(r[1] * r[1] + r[2] * r[2])
          K, sl = Dn / (1 + mu * mu), r[0] - mu * rho
          u1, u2 = r[1] / rho, r[2] / rho
          v = np.array([1.0, -mu * u1, -mu * u2])
          q = -K * mu * sl / rho                       # > 0: curvature of the cone's surface across the sliding direction
          Hc = K * np.outer(v, v)
          Hc[1, 1] += q * (1 - u1 * u1); Hc[1, 2] -= q * u1 * u2; Hc[2, 1] -= q * u1 * u2; Hc[2, 2] += q * (1 - u2 * u2)
          h = Hc @ Jak[c] - K * sl * v
        H += Jc.T @ Hc @ Jc
        g += Jc.T @ h
      a = np.linalg.solve(H, g)
      want = eq | (Ju @ a - aru < 0)
      nfs = np.array([0 if abs(a[j] - ar) <= R * loss else (1 if a[j] - ar > 0 else -1) for (j, ar, R, loss) in fric], int)
      changed = not ((want == act).all() and (nfs == fs).all())
      for c in range(nc):
        Jn_ = J[nr + 3 * c: nr + 3 * c + 3] @ a
        z = self.cone_zone(Jn_ - aref[nr + 3 * c: nr + 3 * c + 3], cone_mu[c])
        if z != zone[c] or (z == 2 and np.abs(Jn_ - Jak[c]).max() > self.ELL_TOL * (1.0 + np.abs(Jak[c]).max())):
          changed = True
        zone[c], Jak[c] = z, Jn_
      act, fs = want, nfs
      if not changed:
        break
    full = np.ones(len(aref), bool); full[:nr] = act
    for c in range(nc):
      full[nr + 3 * c: nr + 3 * c + 3] = zone[c] != 0
    return a, full

  def obb_separated(self, b, bl, pos, quat, cb, Rb, hb):
    """the block's second bounding test: one of the six face axes of (set box, block box) separates them (the margin is part of the set box)"""
    if not hasattr(self, 'col_blk_obb_half'):
      return False
    RA = np.eye(3) if bl < 0 else quat_mat(quat[bl])
    ca = self.col_blk_obb_center[b] if bl < 0 else pos[bl] + RA @ self.col_blk_obb_center[b]
    ha = self.col_blk_obb_half[b]
    R = RA.T @ Rb
    t = RA.T @ (cb - ca)
    aR = np.abs(R)
    for i in range(3):
      if abs(t[i]) > ha[i] + (aR[i, 0] * hb[0] + aR[i, 1] * hb[1] + aR[i, 2] * hb[2]):
        return True
    for j in range(3):
      if abs(t[0] * R[0, j] + t[1] * R[1, j] + t[2] * R[2, j]) > hb[j] + (ha[0] * aR[0, j] + ha[1] * aR[1, j] + ha[2] * aR[2, j]):
        return True
    return False

  def collide(self, pos, quat):
    """sphere / point vs box tests over the model's pair list -> contacts (dist < margin), at most max_contacts"""
    out = []
    near = np.ones(len(self.col_pair), bool)
    if self.block_cull and hasattr(self, 'col_blk_begin'):            # the kernel's block cull, restated so that a wrong bound shows up as a parity failure
      near[:] = False
      for b in range(len(self.col_blk_begin)):
        bl, bi = int(self.col_blk_link[b]), int(self.col_blk_box[b])
        cs = self.col_blk_center[b] if bl < 0 else pos[bl] + quat_mat(quat[bl]) @ self.col_blk_center[b]
        lb = int(self.col_box_link[bi])
        if lb < 0:
          cb, Rb = self.col_box_pos[bi], quat_mat(self.col_box_quat[bi])
        else:
          cb, Rb = pos[lb] + quat_mat(quat[lb]) @ self.col_box_pos[bi], quat_mat(quat_mul(quat[lb], self.col_box_quat[bi]))
        x = Rb.T @ (cs - cb)
        dd = x - np.clip(x, -self.col_box_half[bi], self.col_box_half[bi])
        if (dd ** 2).sum() < self.col_blk_reach[b] ** 2 and not self.obb_separated(b, bl, pos, quat, cb, Rb, self.col_box_half[bi]):
          near[self.col_blk_begin[b]:self.col_blk_end[b]] = True
    blk_of = np.zeros(len(self.col_pair), int)                        # block of every pair, contacts taken per block so far
    for b in range(len(self.col_blk_begin)):
      blk_of[self.col_blk_begin[b]:self.col_blk_end[b]] = b
    caps = getattr(self, 'col_blk_cap', None)
    taken = np.zeros(len(self.col_blk_begin) + 1, int)
    for pi, (si, bi) in enumerate(self.col_pair):
      if not near[pi]:
        continue
      if caps is not None and taken[blk_of[pi]] >= caps[blk_of[pi]]:
        continue
      cls = int(self.col_pair_cls[pi])
      ls, lb = int(self.col_sph_link[si]), int(self.col_box_link[bi])
      c = self.col_sph_pos[si] if ls < 0 else pos[ls] + quat_mat(quat[ls]) @ self.col_sph_pos[si]
      if lb < 0:
        pb, Rb = self.col_box_pos[bi], quat_mat(self.col_box_quat[bi])
      else:
        pb, Rb = pos[lb] + quat_mat(quat[lb]) @ self.col_box_pos[bi], quat_mat(quat_mul(quat[lb], self.col_box_quat[bi]))
      h, r = self.col_box_half[bi], float(self.col_sph_r[si])
      margin = float(self.col_cls_margin[cls])
      if hasattr(self, 'col_box_kind') and self.col_box_kind[bi] == 1:       # edge (segment) vs capsule: closest points of two segments
        ed = self.col_sph_dir[si] if ls < 0 else quat_mat(quat[ls]) @ self.col_sph_dir[si]
        he, cd, hc, rad = float(self.col_sph_hl[si]), Rb[:, 2], float(h[2] - h[0]), float(h[0])
        rr = c - pb
        b_, c_, f_ = ed @ cd, ed @ rr, cd @ rr
        den = 1.0 - b_ * b_
        s_ = min(max((b_ * f_ - c_) / den, -he), he) if den > 1e-12 else 0.0
        t_ = min(max(b_ * s_ + f_, -hc), hc)
        s_ = min(max(b_ * t_ - c_, -he), he)
        pc = pb + t_ * cd
        d = c + s_ * ed - pc
        nd = np.sqrt(d @ d)
        dist = nd - rad
        if dist < margin and nd > 1e-9:
          n = d / nd
          out.append(dict(pair=pi, cls=cls, ls=ls, lb=lb, dist=dist, n=n, p=pc + n * (rad + 0.5 * dist)))
          taken[blk_of[pi]] += 1
          if len(out) == self.max_contacts:
            break
        continue
      x = Rb.T @ (c - pb)
      q = np.clip(x, -h, h)
      if (np.abs(x) > h).any():
        d = x - q
        nd = np.sqrt(d @ d)
        dist, nl = nd - r, d / nd
      else:                                      # centre inside the box: leave through the nearest face
        i = int(np.argmin(h - np.abs(x)))
        sg = 1.0 if x[i] >= 0 else -1.0
        nl = np.zeros(3); nl[i] = sg
        dist = -(h[i] - abs(x[i])) - r
        q = x.copy(); q[i] = sg * h[i]
      if dist < margin:
        n = Rb @ nl
        p = pb + Rb @ q + n * (0.5 * dist)
        out.append(dict(pair=pi, cls=cls, ls=ls, lb=lb, dist=dist, n=n, p=p))
        taken[blk_of[pi]] += 1
        if len(out) == self.max_contacts:
          break
    return out

  @staticmethod
  def tangents(n):
    i = int(np.argmin(np.abs(n)))                # the coordinate axis least aligned with the normal
    e = np.zeros(3); e[i] = 1.0
    t1 = np.cross(n, e); t1 /= np.sqrt(t1 @ t1)
    return t1, np.cross(n, t1)

  def contact_rows(self, contacts, S, qvel):
    """4 pyramid edges per contact: (n +- mu t1, n +- mu t2) . (v_sphere_point - v_box_point); elliptic models: the three rows (n, t1, t2)"""
    nv = self.nv
    J, aref, R = [], [], []
    for c in contacts:
      cls = c['cls']
      mu = float(getattr(self, 'link_mu', {}).get(c['ls'], self.col_cls_mu[cls]))       # link_mu: per-link override (the minitaur's randomised foot friction)
      Jp = np.zeros((3, nv))
      for l, sgn in ((c['ls'], 1.0), (c['lb'], -1.0)):
        if l >= 0:
          for j in self.anc[l]:
            Jp[:, j] += sgn * (S[j, 3:] + np.cross(S[j, :3], c['p']))
      t1, t2 = self.tangents(c['n'])
      margin = float(self.col_cls_margin[cls])
      kk, bb, dd = kbimp(self.col_cls_solref[cls], self.col_cls_solimp[cls], c['dist'] - margin, self.dt)
      R0 = max((1 - dd) / dd * float(self.col_cls_invw[cls]), 1e-15)
      if self.elliptic:                       # (normal, t1, t2), one regulariser (impratio 1); only the normal row has a position term
        for k, d in enumerate((c['n'], t1, t2)):
          row = d @ Jp
          J.append(row)
          aref.append(-bb * (row @ qvel) - (kk * dd * (c['dist'] - margin) if k == 0 else 0.0))
          R.append(R0)
        continue
      for d in (c['n'] + mu * t1, c['n'] - mu * t1, c['n'] + mu * t2, c['n'] - mu * t2):
        row = d @ Jp
        J.append(row)
        aref.append(-bb * (row @ qvel) - kk * dd * (c['dist'] - margin))
        R.append(2 * mu * mu * R0)
    return np.array(J), np.array(aref), np.array(R)

  def step(self, qpos, qvel, ctrl, mocap_pos, mocap_quat, a_prev=None, qfrc=None):
    """a_prev: out['qacc'] of the previous timestep of the same env step (warm start of the active-set iteration), None = cold"""
    out = self.forward(qpos, qvel, ctrl, mocap_pos, mocap_quat, a_prev, qfrc)
    M = out['M']
    qacc = np.linalg.solve(M + self.dt * np.diag(self.jnt_damping), M @ out['qacc'])
    qvel = qvel + self.dt * qacc
    qpos = self.integrate_pos(qpos, qvel)
    return qpos, qvel, out
