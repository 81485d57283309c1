#!/usr/bin/env python3
"""Compile the FACTS of a reference MJCF model into this build's own table format (numeric arrays in an .npz).

Runs only in the build container (the MJCF files live under /root/reference).  What travels is numbers: kinematic tree,
frames, inertias, joint / actuator / equality parameters, collision primitives -- no XML, no meshes.

    python tools/mjcf_compile.py sawyer_door     -> earl_benchmark_amd/models/sawyer_door.npz

Implements the subset of MuJoCo's model compiler these files use (MuJoCo 2.1 semantics, from its documentation):
<include>, nested <default class> + childclass, <compiler angle/inertiafromgeom/inertiagrouprange>, <body> with
pos/quat/euler, <inertial>, <joint> (hinge/slide), <geom> (box/sphere/cylinder/capsule/plane/mesh), <site>,
<position> actuators, <weld> equality, mocap bodies, <option>.  Bodies without <inertial> get mass / inertia from their
geoms in the inertia group range (primitive shapes only; mesh geoms are outside the range in these models).  Mesh geoms
keep their authored frame (MuJoCo re-centres them on the mesh centroid: see `mesh_center` below, needed for the one mesh
geom an observation reads, "handle").
"""
import json
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF_ENVS = '/root/reference/earl_benchmark/envs'
MODELS = {
    'sawyer_door': 'metaworld_assets/sawyer_xyz/sawyer_door_pull.xml',
    'sawyer_peg': 'metaworld_assets/sawyer_xyz/sawyer_peg_insertion_side.xml',
    'kitchen': 'kitchen_assets/adept_envs/adept_envs/franka/assets/franka_kitchen_jntpos_act_ab.xml',
}
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'earl_benchmark_amd', 'models')

# The door panel stands 2.3 cm inside the table top; its four permanent corner contacts reduce to one viscous row on the door hinge
# (oracle/physics_oracle.py, "Permanent deep box-in-box contacts").  With the coefficient derived from the MJCF parameters and the
# calibrated weld (physics_oracle.WELD_TRANSLATION_CALIBRATION) the door moves ~10 % slower than in the MuJoCo recordings: the replayed
# handle ends 3-5 cm behind the recorded one in all five forward demonstrations (path RMS 1.8-2.5 cm).  With 0.8 x the derived value the
# five forward replays follow the recorded handle path within 2-4 mm RMS over the WHOLE episode (finals 3-7 mm; scan 0.6 / 0.75 / 0.8 /
# 0.85 / 1.0: RMS 24-30 / 5-9 / 2-4 / 3-8 / 18-25 mm) and the three reverse replays that keep the rod within 10-13 mm (26-35 mm at 1.0).
# A second CALIBRATION against the recordings (door only), declared as such: which MuJoCo rule accounts for it is not identified
# (candidates: the table's solref 0.02 taken unmixed gives 0.75; saturation of the elliptic cone at the far corners at speed).
# Round 4: with the weld identified on the contact-free prefixes (physics_oracle.WELD_TRANSLATION_CALIBRATION 3.35, rotation 0.07) the factor that
# fits the fit-set door episodes moves to 0.95 (scan 0.8 / 0.85 / 0.9 / 0.95 / 1.0: forward handle-path RMS 14.0 / 8.7 / 3.9 / 2.8 / 6.8 mm) -- within 5 % of the
# derived coefficient; rounds 1 - 3 carried 0.8 next to the weld factor 4.0.
DOOR_DRAG_CALIBRATION = 0.95

# The arm's state at the start of every recorded episode of a task, identified together with the weld factors (tools/weld_free_motion_fit.py --joint,
# profiles/r04_weld_free_motion_fit.json "joint"): seven joint angles and speeds.  It is what the reference's reset recipe -- sim.reset() + 250 timesteps of
# _reset_hand [UPSTREAM], a violent transient from qpos0 with right_j1 outside its range that no stepper but MuJoCo itself reproduces -- leaves behind, as far as the
# recordings show it: the first observation (hand 5.9 / -0.3 / -5.1 mm off the mocap for the door) is met within 0.2 - 0.3 mm, the contact-free hand paths within
# 1 mm RMS.  Written into the link tables as reset_qpos_recorded / reset_qvel_recorded; the envs start from it by default.
RECORDED_RESET = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r04_weld_free_motion_fit.json')))['joint']['tasks']


# ------------------------------------------------------------------ small math
def quat_mul(a, b):
  w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
  return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                   w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quat_norm(q):
  q = np.asarray(q, float)
  return q / np.linalg.norm(q)


def quat_mat(q):
  w, x, y, z = q
  return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def euler_quat(e):  # MuJoCo default eulerseq "xyz": intrinsic rotations about x, then y, then z
  q = np.array([1.0, 0, 0, 0])
  for ang, ax in zip(e, np.eye(3)):
    q = quat_mul(q, np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax]))
  return q


def mat_quat(R):
  t = np.trace(R)
  if t > 0:
    s = np.sqrt(t + 1) * 2
    q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
  else:
    i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1) * 2
    q = [0, 0, 0, 0]
    q[0] = (R[k, j] - R[j, k]) / s; q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
  return quat_norm(q)


def solimp5(s):
  """solimp attribute: missing trailing entries keep MuJoCo's defaults (0.9 0.95 0.001 0.5 2)"""
  d = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
  if s is not None:
    v = np.array([float(x) for x in s.split()], float)
    d[:len(v)] = v
  return d


def vec(s, n=None, default=None):
  if s is None:
    return None if default is None else np.array(default, float)
  v = np.array([float(x) for x in s.split()], float)
  if n is not None and len(v) != n:
    v = np.concatenate([v, np.zeros(n - len(v))])
  return v


# ------------------------------------------------------------------ XML loading (includes + defaults)
def load(path):
  root = ET.parse(path).getroot()
  base = os.path.dirname(path)

  def expand(node):
    out = []
    for ch in list(node):
      if ch.tag == 'include':
        inc = ET.parse(os.path.join(base, ch.get('file'))).getroot()
        expand(inc)
        out.extend(list(inc))
      else:
        expand(ch)
        out.append(ch)
    for ch in list(node):
      node.remove(ch)
    node.extend(out)
  expand(root)
  return root


class Defaults:
  """<default class=...> tree: attribute dicts per element tag, inherited from the parent class."""

  def __init__(self, root):
    self.cls = {'main': {}}
    for d in root.findall('default'):
      self._walk(d, 'main', top=True)

  def _walk(self, node, parent, top=False):
    name = node.get('class') or 'main'
    if name not in self.cls:
      self.cls[name] = {k: dict(v) for k, v in self.cls[parent].items()}
    cur = self.cls[name]
    for ch in node:
      if ch.tag == 'default':
        self._walk(ch, name)
      else:
        cur.setdefault(ch.tag, {}).update(ch.attrib)

  def attrs(self, el, active):
    cls = el.get('class') or active or 'main'
    a = dict(self.cls.get(cls, self.cls['main']).get(el.tag, {}))
    a.update(el.attrib)
    return a


# ------------------------------------------------------------------ inertia of primitives (MuJoCo: uniform density)
def geom_frame(a):
  pos = vec(a.get('pos'), 3, [0, 0, 0])
  if a.get('quat') is not None:
    q = quat_norm(vec(a['quat']))
  elif a.get('euler') is not None:
    q = euler_quat(vec(a['euler']))
  else:
    q = np.array([1.0, 0, 0, 0])
  if a.get('fromto') is not None:
    # capsule / cylinder given by its two end points: centre = midpoint, z axis along the segment, size[1] = half length (set by the caller)
    ft = vec(a['fromto'])
    d = ft[3:] - ft[:3]
    z = d / np.linalg.norm(d)
    x = np.cross([0, 1.0, 0], z) if abs(z[1]) < 0.9 else np.cross([1.0, 0, 0], z)
    x /= np.linalg.norm(x)
    return 0.5 * (ft[:3] + ft[3:]), mat_quat(np.stack([x, np.cross(z, x), z], 1))
  return pos, q


def primitive_inertia(gtype, size, density, mass):
  """mass, principal inertia (about the geom centre, geom axes)"""
  if gtype == 'box':
    x, y, z = size[:3]
    vol = 8 * x * y * z
    m = mass if mass is not None else density * vol
    return m, m / 3 * np.array([y * y + z * z, x * x + z * z, x * x + y * y])
  if gtype == 'sphere':
    r = size[0]
    m = mass if mass is not None else density * 4 / 3 * np.pi * r ** 3
    return m, np.full(3, 0.4 * m * r * r)
  if gtype == 'cylinder':
    r, h = size[0], size[1]
    m = mass if mass is not None else density * np.pi * r * r * 2 * h
    ixy = m * (3 * r * r + (2 * h) ** 2) / 12
    return m, np.array([ixy, ixy, 0.5 * m * r * r])
  if gtype == 'capsule':
    r, h = size[0], size[1]
    vc, vs = np.pi * r * r * 2 * h, 4 / 3 * np.pi * r ** 3
    dens = density if mass is None else mass / (vc + vs)
    mc, ms = dens * vc, dens * vs
    izz = 0.5 * mc * r * r + 0.4 * ms * r * r
    ixx = mc * (3 * r * r + (2 * h) ** 2) / 12 + ms * (0.4 * r * r + 0.375 * r * (2 * h) / 1 * 1 + h * h) if False else \
        mc * (r * r / 4 + (2 * h) ** 2 / 12) + ms * (0.4 * r * r + h * h + 0.75 * r * h)
    return mc + ms, np.array([ixx, ixx, izz])
  raise NotImplementedError(gtype)


def combine_inertia(parts):
  """parts: (mass, com[3], R[3,3] of principal axes, diag[3]) in the body frame -> mass, com, quat, diag"""
  M = sum(p[0] for p in parts)
  if M <= 0:
    return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
  com = sum(p[0] * p[1] for p in parts) / M
  I = np.zeros((3, 3))
  for m, c, R, d in parts:
    I += R @ np.diag(d) @ R.T
    r = c - com
    I += m * (r @ r * np.eye(3) - np.outer(r, r))
  w, V = np.linalg.eigh(I)
  order = np.argsort(-w)           # MuJoCo sorts principal inertias in decreasing order
  w, V = w[order], V[:, order]
  if np.linalg.det(V) < 0:
    V[:, 2] = -V[:, 2]
  return M, com, mat_quat(V), w


# ------------------------------------------------------------------ STL centroid (MuJoCo re-centres mesh geoms on it)
def mesh_center(path):
  """Centre MuJoCo 2.1 gives a mesh: volume-weighted centroid of the pyramids spanned by each face and the mean of the
  face centroids (|volume| weights), see mesh processing in the MuJoCo docs ("Mesh")."""
  data = open(path, 'rb').read()
  n = struct.unpack('<I', data[80:84])[0]
  tri = np.frombuffer(data, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]), count=n, offset=84)['v'].astype(float)
  fc = tri.mean(1)
  area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
  c0 = (fc * area[:, None]).sum(0) / area.sum()
  a, b, c = tri[:, 0] - c0, tri[:, 1] - c0, tri[:, 2] - c0
  vol = np.abs(np.einsum('ij,ij->i', a, np.cross(b, c))) / 6
  cen = (tri.sum(1) + c0) / 4
  return (cen * vol[:, None]).sum(0) / vol.sum(), c0, tri


def mesh_inertia(path, scale, mass):
  """mass properties of a mesh geom with a given mass: the mesh is the union of the pyramids spanned by its faces and the centre c0
  (MuJoCo 2.1 processes meshes this way, with |volume| weights: exact for shapes that are star-convex about c0, which these collision
  hulls are) -> com (mesh frame), inertia tensor about the com (mesh frame)"""
  _, c0, tri = mesh_center(path)
  tri = tri * np.asarray(scale, float)
  c0 = c0 * np.asarray(scale, float)
  a, b, c = tri[:, 0] - c0, tri[:, 1] - c0, tri[:, 2] - c0
  vol = np.abs(np.einsum('ij,ij->i', a, np.cross(b, c))) / 6
  cen = (a + b + c) / 4                              # tetrahedron centroid relative to c0 (fourth vertex at the origin)
  V = vol.sum()
  com_rel = (cen * vol[:, None]).sum(0) / V
  # second moments of a tetrahedron with one vertex at the origin: integral x x' dV = V / 20 * (sum_i v_i v_i' + (sum v)(sum v)')
  S = np.zeros((3, 3))
  for k in range(len(tri)):
    vs = np.stack([a[k], b[k], c[k]])
    sm = vs.sum(0)
    S += vol[k] / 20.0 * (vs.T @ vs + np.outer(sm, sm))
  dens = mass / V
  C = dens * S                                       # about c0
  I0 = np.trace(C) * np.eye(3) - C
  I = I0 - mass * (com_rel @ com_rel * np.eye(3) - np.outer(com_rel, com_rel))
  return c0 + com_rel, I


# ------------------------------------------------------------------ the compiler
def compile_model(name, path=None):
  """path: an MJCF file to read instead of the reference's (tests/test_pin_tool.py feeds the text tools/pin_with_simulator.py emits from this build's own
  tables back through this compiler)"""
  path = path or os.path.join(REF_ENVS, MODELS[name])
  root = load(path)
  dfl = Defaults(root)
  comp = {}
  for c in root.findall('compiler'):
    comp.update(c.attrib)
  assert comp.get('angle', 'degree') == 'radian'
  grp_lo, grp_hi = [int(x) for x in comp.get('inertiagrouprange', '0 5').split()]
  opt = {}
  for o in root.findall('option'):
    opt.update(o.attrib)
  meshes, mesh_scale = {}, {}
  meshdir = os.path.normpath(os.path.join(os.path.dirname(path), comp.get('meshdir', '.')))
  for a in root.findall('asset'):
    for m in a.findall('mesh'):
      meshes[m.get('name')] = os.path.normpath(os.path.join(meshdir, m.get('file')))
      mesh_scale[m.get('name')] = vec(m.get('scale'), 3, [1, 1, 1])

  bodies, joints, geoms, sites = [], [], [], []
  names = {'body': ['world'], 'joint': [], 'geom': [], 'site': []}
  bodies.append(dict(parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), mocap=0, inertial=None, geoms=[]))

  def add_geom(el, bid, active):
    a = dfl.attrs(el, active)
    gtype = a.get('type', 'sphere')
    pos, q = geom_frame(a)
    size = vec(a.get('size'), 3, [0, 0, 0])
    if a.get('fromto') is not None:
      ft = vec(a['fromto'])
      size = np.array([size[0], 0.5 * np.linalg.norm(ft[3:] - ft[:3]), 0.0])
    g = dict(body=bid, type=gtype, pos=pos, quat=q, size=size, group=int(a.get('group', 0)),
             contype=int(a.get('contype', 1)), conaffinity=int(a.get('conaffinity', 1)), condim=int(a.get('condim', 3)),
             density=float(a.get('density', 1000)), mass=(float(a['mass']) if 'mass' in a else None),
             friction=vec(a.get('friction'), 3, [1, 0.005, 0.0001]), solref=vec(a.get('solref'), 2, [0.02, 1]),
             solimp=vec(a.get('solimp'), 5, [0.9, 0.95, 0.001, 0.5, 2]), margin=float(a.get('margin', 0)),
             mesh=a.get('mesh'))
    inertial_mesh = gtype == 'mesh' and grp_lo <= g['group'] <= grp_hi and g['mass'] is not None and g['mass'] > 0
    if gtype == 'mesh' and (inertial_mesh or name != 'kitchen'):
      # MuJoCo moves the geom frame to the mesh centre (orientation: principal axes, not needed here).  (The kitchen's purely visual
      # meshes -- group 0 / 1, no mass, no collision -- are never read by anything and are left where they are.)
      cen, _, _ = mesh_center(meshes[a['mesh']])
      g['pos'] = pos + quat_mat(q) @ (cen * mesh_scale[a['mesh']])
    if inertial_mesh:
      com, I = mesh_inertia(meshes[a['mesh']], mesh_scale[a['mesh']], g['mass'])
      g['mesh_com'], g['mesh_I'] = pos + quat_mat(q) @ com, quat_mat(q) @ I @ quat_mat(q).T      # body frame
    names['geom'].append(a.get('name', ''))
    geoms.append(g)
    bodies[bid]['geoms'].append(len(geoms) - 1)

  def walk(el, parent, active):
    for ch in el:
      if ch.tag == 'body':
        cls = ch.get('childclass') or active
        pos = vec(ch.get('pos'), 3, [0, 0, 0])
        q = quat_norm(vec(ch.get('quat'))) if ch.get('quat') else (euler_quat(vec(ch.get('euler'))) if ch.get('euler') else np.array([1.0, 0, 0, 0]))
        bid = len(bodies)
        bodies.append(dict(parent=parent, pos=pos, quat=q, mocap=int(ch.get('mocap', 'false') == 'true'), inertial=None, geoms=[]))
        names['body'].append(ch.get('name', ''))
        for sub in ch:
          if sub.tag == 'inertial':
            iq = quat_norm(vec(sub.get('quat'))) if sub.get('quat') else np.array([1.0, 0, 0, 0])
            bodies[bid]['inertial'] = (float(sub.get('mass')), vec(sub.get('pos'), 3), iq, vec(sub.get('diaginertia'), 3))
          elif sub.tag == 'joint':
            a = dfl.attrs(sub, cls)
            base = dict(body=bid, pos=vec(a.get('pos'), 3, [0, 0, 0]), limited=int(a.get('limited', 'false') == 'true'),
                        range=vec(a.get('range'), 2, [0, 0]), damping=float(a.get('damping', 0)),
                        armature=float(a.get('armature', 0)), frictionloss=float(a.get('frictionloss', 0)), stiffness=float(a.get('stiffness', 0)),
                        springref=float(a.get('springref', 0)), solref=vec(a.get('solreflimit'), 2, [0.02, 1]),
                        solimp=vec(a.get('solimplimit'), 5, [0.9, 0.95, 0.001, 0.5, 2]), margin=float(a.get('margin', 0)))
            if a.get('type', 'hinge') == 'free':
              # A free joint becomes six 1-dof entries on the same body, in MuJoCo's dof order: three translations along the
              # WORLD axes ('slide'), then the rotation about the three BODY axes: 'ball0' carries the orientation quaternion
              # (qpos keeps MuJoCo's layout: xyz + wxyz), 'ball1' / 'ball2' only contribute their axis.  The body's own pos /
              # quat are the joint's qpos0 (MuJoCo semantics), so the body frame itself is zeroed.
              assert parent == 0 and not base['limited']
              bodies[bid]['free'] = 1
              bodies[bid]['qpos0'] = np.concatenate([bodies[bid]['pos'], bodies[bid]['quat']])
              bodies[bid]['pos'], bodies[bid]['quat'] = np.zeros(3), np.array([1.0, 0, 0, 0])
              for k, t in enumerate(('slide', 'slide', 'slide', 'ball0', 'ball1', 'ball2')):
                joints.append(dict(base, type=t, axis=np.eye(3)[k % 3], pos=np.zeros(3)))
                names['joint'].append(a.get('name', names['body'][bid]) + '_' + 'xyz'[k % 3] + ('t' if k < 3 else 'r'))
            else:
              ax = vec(a.get('axis'), 3, [0, 0, 1])
              joints.append(dict(base, type=a.get('type', 'hinge'), axis=ax / np.linalg.norm(ax)))
              names['joint'].append(a.get('name', ''))
          elif sub.tag == 'geom':
            add_geom(sub, bid, cls)
          elif sub.tag == 'site':
            a = dfl.attrs(sub, cls)
            pos, q = geom_frame(a)
            sites.append(dict(body=bid, pos=pos, quat=q))
            names['site'].append(a.get('name', ''))
        walk(ch, bid, cls)
      elif ch.tag == 'geom' and el.tag == 'worldbody':
        add_geom(ch, 0, active)
      elif ch.tag == 'site' and el.tag == 'worldbody':
        a = dfl.attrs(ch, active)
        pos, q = geom_frame(a)
        sites.append(dict(body=0, pos=pos, quat=q))
        names['site'].append(a.get('name', ''))

  for wb in root.findall('worldbody'):
    walk(wb, 0, None)

  # inertia: explicit <inertial>, else from geoms in the inertia group range
  nb = len(bodies)
  mass, ipos, iquat, inertia = np.zeros(nb), np.zeros((nb, 3)), np.tile([1.0, 0, 0, 0], (nb, 1)), np.zeros((nb, 3))
  for b, bd in enumerate(bodies):
    if bd['inertial'] is not None:
      mass[b], ipos[b], iquat[b], inertia[b] = bd['inertial']
    else:
      parts = []
      for gi in bd['geoms']:
        g = geoms[gi]
        if grp_lo <= g['group'] <= grp_hi and g['type'] not in ('plane', 'mesh'):
          m, d = primitive_inertia(g['type'], g['size'], g['density'], g['mass'])
          parts.append((m, g['pos'], quat_mat(g['quat']), d))
        elif 'mesh_I' in g:                       # mesh geom with a given mass (the Franka links' collision hulls)
          w, V = np.linalg.eigh(g['mesh_I'])
          parts.append((g['mass'], g['mesh_com'], V, w))
      if parts:
        mass[b], ipos[b], iquat[b], inertia[b] = combine_inertia(parts)

  jtype = {'hinge': 0, 'slide': 1, 'ball0': 2, 'ball1': 3, 'ball2': 3}
  gtype = {'plane': 0, 'sphere': 1, 'capsule': 2, 'cylinder': 3, 'box': 4, 'mesh': 5}
  acts = []
  for ac in root.findall('actuator'):
    for el in ac:
      a = dfl.attrs(el, None)
      assert el.tag == 'position'
      a = dfl.attrs(el, el.get('class'))
      acts.append(dict(joint=names['joint'].index(a['joint']), kp=float(a.get('kp', 1)), ctrlrange=vec(a.get('ctrlrange'), 2, [0, 0]),
                       ctrllimited=int(a.get('ctrllimited', 'false') == 'true'),
                       forcerange=vec(a.get('forcerange'), 2, [0, 0]) if a.get('forcelimited', 'false') == 'true' else np.array([-np.inf, np.inf])))
  welds, jeqs = [], []
  for eq in root.findall('equality'):
    for el in eq:
      if el.tag == 'joint':                          # joint1 - ref1 = polynomial(joint2 - ref2): these models use the linear term only
        pc = vec(el.get('polycoef'), 5, [0, 1, 0, 0, 0])
        assert pc[2] == pc[3] == pc[4] == 0
        jeqs.append(dict(joint1=names['joint'].index(el.get('joint1')), joint2=names['joint'].index(el.get('joint2')), c0=pc[0], c1=pc[1],
                         solref=vec(el.get('solref'), 2, [0.02, 1]), solimp=solimp5(el.get('solimp'))))
        continue
      assert el.tag == 'weld'
      welds.append(dict(body1=names['body'].index(el.get('body1')), body2=names['body'].index(el.get('body2')),
                        solref=vec(el.get('solref'), 2, [0.02, 1]), solimp=solimp5(el.get('solimp'))))

  out = dict(
      name=np.array(name), timestep=np.float64(opt.get('timestep', 0.002)), cone_elliptic=np.int32(opt.get('cone', 'pyramidal') == 'elliptic'), gravity=vec(opt.get('gravity'), 3, [0, 0, -9.81]),
      body_parent=np.array([b['parent'] for b in bodies], np.int32), body_pos=np.stack([b['pos'] for b in bodies]),
      body_quat=np.stack([b['quat'] for b in bodies]), body_mocap=np.array([b['mocap'] for b in bodies], np.int32),
      body_free=np.array([b.get('free', 0) for b in bodies], np.int32),
      body_qpos0=np.stack([b.get('qpos0', np.array([0, 0, 0, 1.0, 0, 0, 0])) for b in bodies]),
      body_mass=mass, body_ipos=ipos, body_iquat=iquat, body_inertia=inertia,
      jnt_body=np.array([j['body'] for j in joints], np.int32), jnt_type=np.array([jtype[j['type']] for j in joints], np.int32),
      jnt_axis=np.stack([j['axis'] for j in joints]), jnt_pos=np.stack([j['pos'] for j in joints]),
      jnt_limited=np.array([j['limited'] for j in joints], np.int32), jnt_range=np.stack([j['range'] for j in joints]),
      jnt_damping=np.array([j['damping'] for j in joints]), jnt_armature=np.array([j['armature'] for j in joints]),
      jnt_solref=np.stack([j['solref'] for j in joints]), jnt_solimp=np.stack([j['solimp'] for j in joints]),
      geom_body=np.array([g['body'] for g in geoms], np.int32), geom_type=np.array([gtype[g['type']] for g in geoms], np.int32),
      geom_pos=np.stack([g['pos'] for g in geoms]), geom_quat=np.stack([g['quat'] for g in geoms]),
      geom_size=np.stack([g['size'] for g in geoms]), geom_contype=np.array([g['contype'] for g in geoms], np.int32),
      geom_conaffinity=np.array([g['conaffinity'] for g in geoms], np.int32), geom_condim=np.array([g['condim'] for g in geoms], np.int32),
      geom_friction=np.stack([g['friction'] for g in geoms]), geom_solref=np.stack([g['solref'] for g in geoms]),
      geom_solimp=np.stack([g['solimp'] for g in geoms]), geom_margin=np.array([g['margin'] for g in geoms]),
      site_body=np.array([s['body'] for s in sites], np.int32), site_pos=np.stack([s['pos'] for s in sites]),
      act_joint=np.array([a['joint'] for a in acts], np.int32), act_kp=np.array([a['kp'] for a in acts]),
      act_ctrlrange=np.stack([a['ctrlrange'] for a in acts]),
      weld_body1=np.array([w['body1'] for w in welds], np.int32), weld_body2=np.array([w['body2'] for w in welds], np.int32),
      weld_solref=np.stack([w['solref'] for w in welds]), weld_solimp=np.stack([w['solimp'] for w in welds]),
      jnt_frictionloss=np.array([j['frictionloss'] for j in joints]), jnt_stiffness=np.array([j['stiffness'] for j in joints]),
      jnt_springref=np.array([j['springref'] for j in joints]), act_forcerange=np.stack([a['forcerange'] for a in acts]),
      jeq_joint1=np.array([e['joint1'] for e in jeqs], np.int32), jeq_joint2=np.array([e['joint2'] for e in jeqs], np.int32),
      jeq_coef=np.array([[e['c0'], e['c1']] for e in jeqs], float).reshape(-1, 2),
      jeq_solref=np.array([e['solref'] for e in jeqs], float).reshape(-1, 2), jeq_solimp=np.array([e['solimp'] for e in jeqs], float).reshape(-1, 5),
      key_qpos=np.array([float(x) for k in root.findall('keyframe') for kk in k.findall('key') for x in kk.get('qpos', '').split()]),
      body_names=np.array(names['body']), joint_names=np.array(names['joint']), geom_names=np.array(names['geom']),
      site_names=np.array(names['site']),
  )
  return out


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else 'sawyer_door'
  m = compile_model(name)
  # constants MuJoCo's compiler derives at qpos0: inverse inertia weights used by the constraint regularizer
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from oracle import physics_oracle as po
  bw, dw = po.inverse_weights(po.Model(m))
  m['body_invweight0'], m['dof_invweight0'] = bw, dw
  if name in ('sawyer_door', 'sawyer_peg'):      # the models with recordings: round 4's factors; the others keep the round-2 pair (Model's default, LEGACY_WELD_CALIBRATION)
    m['weld_calibration'] = np.array([po.WELD_TRANSLATION_CALIBRATION, po.WELD_ROTATION_CALIBRATION])
  os.makedirs(OUT, exist_ok=True)
  np.savez_compressed(os.path.join(OUT, name + '.npz'), **m)
  if name == 'sawyer_door':
    # link form consumed by the HIP stepper; the env moves the door body to obj_init_pos at construction
    # (envs/sawyer_door.py:36, :111-113) -- baked in here
    pm = po.Model(m)
    bp = pm.body_pos.copy()
    bp[pm.body_id('door')] = np.array([0.1, 0.95, 0.1], np.float32).astype(float)
    # collision set of the door task: gripper plates vs handle bars (cylinders on door_link), plate corners vs the door
    # panel, the door frame and the table top.  Left out: the two mesh hulls on the wrist (l6, eGripperBase), walls, floor.
    dl, frame, table = pm.body_id('door_link'), pm.body_id('doorlockB'), pm.body_id('tablelink')
    colliding = lambda g: bool(m['geom_contype'][g] or m['geom_conaffinity'][g])
    chains = [g for g in range(len(m['geom_body'])) if m['geom_body'][g] == dl and m['geom_type'][g] == 3 and colliding(g)
              and m['geom_size'][g][1] > m['geom_size'][g][0]]
    big = [g for g in range(len(m['geom_body'])) if m['geom_body'][g] in (dl, frame, table) and m['geom_type'][g] == 4 and colliding(g)]
    panel = [g for g in big if m['geom_body'][g] == dl][0]
    tbl = [g for g in big if m['geom_body'][g] == table][0]
    # DOOR_CONTACTS (environment, experiments of round 5 / VERDICT r04 item 1): 'chains' = rounds 1 - 4 (sphere chains + edge caps on the handle's four cylinders, one
    # merged plate per finger); 'cyl' = the four cylinders as cylinders, one contact per (box, cylinder) pair; '+split' = claw plate and pad as two boxes with their own
    # contact parameters; '+tor' = condim-4 torsional coefficient in the contact classes.  Output: sawyer_door_links.npz, or sawyer_door_<variant>_links.npz.
    variant = os.environ.get('DOOR_CONTACTS', 'chains')
    allcyl = [g for g in range(len(m['geom_body'])) if m['geom_body'][g] == dl and m['geom_type'][g] == 3 and colliding(g)]
    if variant.startswith('cyl'):
      fingers = [['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']]
      red = po.reduce_model(pm, bp, attach_bodies=['hand'], attach_sites=['rightEndEffector', 'leftEndEffector', 'endEffector'], attach_geoms=['handle'],
                            collision=dict(plates=fingers, plates_split='split' in variant, plates_accept=('cyl',), torsion='tor' in variant,
                                           cylinders=[dict(geom=g) for g in allcyl], corner_sets=fingers, big_boxes=big,
                                           set_priority=('cyl',), set_cap=dict(cyl=4),
                                           drag=[(panel, tbl, -(bp[pm.body_id('door')][2] - m['geom_size'][panel][2]))], drag_calibration=DOOR_DRAG_CALIBRATION))
      red['reset_qpos_recorded'], red['reset_qvel_recorded'] = (np.array(RECORDED_RESET[name][k]) for k in ('start_qpos', 'start_qvel'))
      np.savez_compressed(os.path.join(OUT, f'{name}_{variant.replace("+", "_")}_links.npz'), **red)
      print('door variant', variant, 'spheres', len(red['col_sph_link']), 'boxes', len(red['col_box_link']), 'pairs', len(red['col_pair']), 'blocks', len(red['col_blk_begin']),
            'classes', len(red['col_cls_mu']), 'mu_tor', red['col_cls_mu_tor'])
      return
    red = po.reduce_model(pm, bp, attach_bodies=['hand'], attach_sites=['rightEndEffector', 'leftEndEffector', 'endEffector'],
                          attach_geoms=['handle'],
                          collision=dict(plates=[['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']],
                                         chains=[dict(geom=g, spacing=0.9 if m['geom_size'][g][0] < 0.015 else 0.75) for g in chains],   # 16 spheres: one pass per plate
                                        
                                         corner_sets=[['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']], big_boxes=big,
                                         edge_caps=dict(plates=[['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']], caps=chains, per_plate=True),
                                         set_priority=('edge0', 'edge1'), set_cap=dict(edge0=1, edge1=1), drop_contained=True,
                                         drag=[(panel, tbl, -(bp[pm.body_id('door')][2] - m['geom_size'][panel][2]))],
                                         drag_calibration=DOOR_DRAG_CALIBRATION))
    red['reset_qpos_recorded'], red['reset_qvel_recorded'] = (np.array(RECORDED_RESET[name][k]) for k in ('start_qpos', 'start_qvel'))
    np.savez_compressed(os.path.join(OUT, name + '_links.npz'), **red)
  if name == 'sawyer_peg':
    # link form of the peg task.  reset_model puts body 'box' at goal - (0.03, 0, 0.13) = its MJCF position (sawyer_peg.py:196-197).
    # Collision set: the peg (a 3 x 3 x 24 cm box) is a chain of inscribed spheres against the gripper plates and the
    # seven boxes of the hole block, and its 8 corners are points against the table top and the block; the plate corners
    # are points against the table top, the block and the four retaining walls.  Left out: wrist mesh hulls, floor.
    pm = po.Model(m)
    table = pm.body_id('tablelink')
    blockb = [b for b in range(len(m['body_parent'])) if m['body_parent'][b] == pm.body_id('box')][0]
    colliding = lambda g: bool(m['geom_contype'][g] or m['geom_conaffinity'][g])
    boxes = lambda body: [g for g in range(len(m['geom_body'])) if m['geom_body'][g] == body and m['geom_type'][g] == 4 and colliding(g)]
    peg = pm.geom_id('peg')
    red = po.reduce_model(pm, None, attach_bodies=['hand', 'leftpad', 'rightpad'],
                          attach_sites=['rightEndEffector', 'leftEndEffector', 'endEffector', 'pegHead', 'pegGrasp'],
                          collision=dict(max_contacts=12, set_priority=('pegcorner', 'peg', 'corner'), set_cap=dict(pegcorner=4, peg=4, corner=2), plates=[['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']], plates_accept=('peg',),
                                         chains=[dict(geom=peg, set='peg', spacing=1.0)],
                                         corner_sets=[dict(names=['peg'], set='pegcorner'), ['rightclaw_it', 'rightpad_geom'], ['leftclaw_it', 'leftpad_geom']],
                                         big_boxes=[dict(geom=g, accept=('peg', 'pegcorner', 'corner')) for g in boxes(blockb)] +
                                                   [dict(geom=g, accept=('pegcorner', 'corner')) for g in boxes(table)] +
                                                   # the four retaining walls at the table's edges (scene/basic_scene.xml:49-58) keep a pushed peg on the
                                                   # table (round 2; round 1 left them out: "a peg pushed off the table would fall forever")
                                                   [dict(geom=g, accept=('pegcorner',)) for g in boxes(pm.body_id('RetainingWall'))]))
    # world-fixed sites the dense reward reads (sawyer_peg.py:252-256): the corners of the two keep-out prisms in front of the hole block
    kin0 = po.kinematics(pm, po.dof_qpos0(pm))
    corner = lambda n: kin0['xpos'][pm.site_body[pm.site_id(n)]] + kin0['xmat'][pm.site_body[pm.site_id(n)]] @ pm.site_pos[pm.site_id(n)]
    red['peg_box_corners'] = np.stack([corner(n) for n in ('bottom_right_corner_collision_box_1', 'top_left_corner_collision_box_1',
                                                            'bottom_right_corner_collision_box_2', 'top_left_corner_collision_box_2')])
    red['reset_qpos_recorded'], red['reset_qvel_recorded'] = (np.array(RECORDED_RESET[name][k]) for k in ('start_qpos', 'start_qvel'))
    np.savez_compressed(os.path.join(OUT, name + '_links.npz'), **red)
    print('links:', len(red['parent']), 'spheres', len(red['col_sph_link']), 'boxes', len(red['col_box_link']), 'pairs', len(red['col_pair']),
          'blocks', len(red['col_blk_begin']), 'classes', len(red['col_cls_mu']))
  if name == 'kitchen':
    # Link form of the kitchen (BASELINE configs[3]): nv = 23 = 7 arm hinges + 2 finger slides + 14 single-dof fixtures
    # (each its own tree), 5 joint couplings, dry friction on six joints, the mocap weld on panda0_link7.
    # Collision set (this build's own, DECLARED: MuJoCo's 117 colliding geoms incl. nine convex link hulls are not reproduced): each finger
    # (twelve capsules + a box) is ONE box; the graspable fixtures are sphere chains -- microwave handle, the two hinge-door handles, the
    # slide-door handle, the four knob bars, the light switch -- tested against the two finger boxes; the finger boxes' corners are points
    # tested against the four door panels.  Left out: counters, oven body, walls, floor, the arm links' hulls, the handle stubs, the knob discs.
    pm = po.Model(m)
    gb, gt, gs = m['geom_body'], m['geom_type'], m['geom_size']
    colliding = lambda g: bool(m['geom_contype'][g] or m['geom_conaffinity'][g])
    on = lambda body: [g for g in range(len(gb)) if gb[g] == pm.body_id(body) and colliding(g)]
    longest = lambda gl, types: max((g for g in gl if gt[g] in types), key=lambda g: gs[g][1] if gt[g] in (2, 3) else max(gs[g]))
    chains = [dict(geom=longest(on('microdoorroot'), (2, 3)), set='micro', spacing=1.5), dict(geom=longest(on('hingeleftdoor'), (2, 3)), set='hingel', spacing=1.5),
              dict(geom=longest(on('hingerightdoor'), (2, 3)), set='hinger', spacing=1.5), dict(geom=longest(on('slidelink'), (2, 3)), set='slide', spacing=1.5),
              dict(geom=longest(on('lightswitchroot'), (2, 3)), set='light', spacing=1.5)]
    for k in range(1, 5):
      chains.append(dict(geom=[g for g in on(f'knob {k}') if gt[g] == 4][0], set=f'knob{k}', spacing=1.5))
    panels = [[g for g in on(b) if gt[g] == 4][0] for b in ('microdoorroot', 'hingeleftdoor', 'hingerightdoor', 'slidelink')]
    finger_geom = {b: [g for g in on(b) if gt[g] == 4][0] for b in ('panda0_leftfinger', 'panda0_rightfinger')}      # contact parameters of the finger class
    sets = tuple(c['set'] for c in chains)
    fingers = [dict(body='panda0_leftfinger', pos=[0.0, 0.0125, 0.0475], half=[0.0185, 0.0125, 0.0475], like=finger_geom['panda0_leftfinger'], accept=sets, corners='tipl'),
               dict(body='panda0_rightfinger', pos=[0.0, -0.0125, 0.0475], half=[0.0185, 0.0125, 0.0475], like=finger_geom['panda0_rightfinger'], accept=sets, corners='tipr')]
    # Round 3: the HAND against the kitchen's big static boxes.  The hand (link 7: the wrist flange, the hand's hull and the
    # envelope the two fingers sweep) is stood in for by eight spheres -- this build's own fit to the panda_col hulls' extents (flange r 0.05 at
    # z 0.08; hand body 0.2 x 0.065 x 0.06 centred at z 0.131 as three spheres r 0.035 along the fingers' slide axis; finger envelope to the tips at
    # z 0.26 as four spheres r 0.02) -- tested against SIX static boxes taken from the MJCF's own collision geoms: the counter-top slab, the oven /
    # stove body, the back wall, the microwave's body (hull of its six plates), the bottom plates of the slide and hinge cabinets (as one box) and the
    # hood; likewise the wrist (link 6), the forearm (link 5) and the finger boxes' corner points.  Links 1-4, the right counter and the floor stay uncollided.
    ARM_SETS = ('hand', 'wrist', 'forearm', 'tipl', 'tipr')      # what meets the static boxes: three links' sphere sets and the two fingers' corner points
    world_box = lambda body, sel: [g for g in on(body) if gt[g] == 4 and sel(g)]
    kin0 = po.kinematics(pm, np.zeros(len(m['jnt_body'])))
    gcen = lambda g: kin0['xpos'][gb[g]] + kin0['xmat'][gb[g]] @ m['geom_pos'][g]
    pick = lambda body, centre: min(world_box(body, lambda g: True), key=lambda g: float(np.abs(gcen(g) - np.asarray(centre)).sum()))
    statics = [pick('counters', (-0.955, 0.242, 1.57)), pick('ovenroot', (0.0, 0.738, 0.808)), pick('wallroot', (-0.151, 1.394, 2.187)), pick('hoodroot', (0.0, 1.011, 2.284))]

    def hull(geoms):                                   # axis-aligned hull of world-fixed boxes (they are all unrotated) -> explicit box in the world frame
      lo = np.min([gcen(g) - gs[g][:3] for g in geoms], 0); hi = np.max([gcen(g) + gs[g][:3] for g in geoms], 0)
      return dict(body='world', pos=0.5 * (lo + hi), half=0.5 * (hi - lo), like=geoms[0], accept=ARM_SETS, late=True)
    micro_body = hull(world_box('microroot', lambda g: True))
    cab_bottoms = hull(world_box('slide', lambda g: abs(gcen(g)[2] - 2.42) < 0.01) + world_box('hingecab', lambda g: abs(gcen(g)[2] - 2.42) < 0.01))
    # Round 4: the RIGHT-hand counter -- its body and the eight slab / rim pieces around the sink (adept_models/kitchen/assets/counters_asset.xml
    # through counters_chain.xml; world x 0.50 .. 1.11, y 0.09 .. 1.32, top at z 1.60) as ONE box, the hull of the nine (the sink's hole is filled: declared).
    # Appended after the round-3 set, whose pairs and blocks keep their places.  What stays uncollided, with the reason: the FLOOR (z = 0) and the left counter's
    # body under its slab lie outside the arm's reach (base at z = 1.80, shoulder at 2.13, reach 0.86 + 0.2 m of hand: nothing of the arm gets below z = 1.07, and
    # the slab overhangs the body); links 1-4 (shoulder 2.13, elbow within 0.40 m of it) stay >= 0.2 m clear of every box of the scene in every pose.
    right_counter = hull(world_box('counters', lambda g: gcen(g)[0] > 0.4))
    yf = np.array([np.sin(np.pi / 4), np.cos(np.pi / 4), 0.0])       # the fingers' slide axis in the frame of link 7 (the finger frames are turned -45 deg about z)
    hand = [dict(body='panda0_link7', pos=[0, 0, 0.08], r=0.05, set='hand', like=finger_geom['panda0_leftfinger'])]
    hand += [dict(body='panda0_link7', pos=list(t * yf + [0, 0, 0.131]), r=0.035, set='hand', like=finger_geom['panda0_leftfinger']) for t in (-0.07, 0.0, 0.07)]
    hand += [dict(body='panda0_link7', pos=list(t * yf + [0, 0, z]), r=0.02, set='hand', like=finger_geom['panda0_leftfinger']) for t in (-0.04, 0.04) for z in (0.21, 0.245)]
    # ... and the two links above it, fitted to the extents of their collision hulls (third_party/franka/meshes/collision/link5.stl: x +-0.055, y -0.055 .. 0.13,
    # z -0.265 .. 0.052; link6.stl: x -0.048 .. 0.132, y -0.051 .. 0.082, z +-0.05): the forearm as three spheres r 0.06 along its length, the wrist as two r 0.055.
    # (Links 1-4 cannot reach any of the six boxes and stay uncollided.)
    hand += [dict(body='panda0_link5', pos=[0.0, 0.02, z], r=0.06, set='forearm', like=finger_geom['panda0_leftfinger']) for z in (-0.21, -0.105, 0.0)]
    hand += [dict(body='panda0_link6', pos=p_, r=0.055, set='wrist', like=finger_geom['panda0_leftfinger']) for p_ in ([0.0, 0.01, 0.0], [0.085, 0.02, 0.0])]
    red = po.reduce_model(pm, None, attach_bodies=['panda0_link7'],
                          attach_sites=['end_effector', 'knob1_site', 'knob2_site', 'knob3_site', 'knob4_site', 'light_site', 'slide_site', 'hinge_site2', 'microhandle_site'],
                          weld_translation_calibration=1.0,        # no recordings of this env exist: the derived value, not the Sawyer calibration
                          collision=dict(max_contacts=12, explicit_boxes=fingers + [micro_body, cab_bottoms, right_counter], chains=chains, explicit_spheres=hand,
                                         big_boxes=[dict(geom=g, accept=('tipl', 'tipr')) for g in panels] + [dict(geom=g, accept=ARM_SETS) for g in statics],
                                         set_priority=sets, set_cap=dict({c['set']: 4 for c in chains}, tipl=2, tipr=2, hand=3, wrist=2, forearm=2)))
    red['key_qpos'] = m['key_qpos']
    np.savez_compressed(os.path.join(OUT, name + '_links.npz'), **red)
    print('links:', len(red['parent']), 'spheres', len(red['col_sph_link']), 'boxes', len(red['col_box_link']), 'pairs', len(red['col_pair']),
          'blocks', len(red['col_blk_begin']), 'classes', len(red['col_cls_mu']), 'attachments', list(red['att_names']))
  nb, nj, ng = len(m['body_parent']), len(m['jnt_body']), len(m['geom_body'])
  col = int(((m['geom_contype'] != 0) | (m['geom_conaffinity'] != 0)).sum())
  print(f'{name}: {nb} bodies, {nj} joints (nv={nj}), {ng} geoms ({col} colliding), {len(m["act_joint"])} actuators, '
        f'{len(m["weld_body1"])} welds, dt={float(m["timestep"])}, total mass {m["body_mass"].sum():.3f}')
  for j in range(nj):
    print(f'  joint {j} {m["joint_names"][j]:12s} body {m["body_names"][m["jnt_body"][j]]:12s} type {m["jnt_type"][j]} '
          f'range {m["jnt_range"][j]} damping {m["jnt_damping"][j]} armature {m["jnt_armature"][j]}')


if __name__ == '__main__':
  main()
