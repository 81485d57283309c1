#!/usr/bin/env python3
"""Pin the HIP stepper against a REAL simulator, if one is importable on this host.

The reference binds MuJoCo 2.1 through mujoco-py (env.yml:11, envs/kitchen_assets/adept_envs/adept_envs/simulation/sim_robot.py:67-69) and PyBullet 3.2
(env.yml:14, envs/minitaur_gym_env.py:16-22).  Neither ships with this image, so every rigid-body row of DESIGN.md is "parity unpinned".  This tool is
the route out of that: it

  1. probes for `mujoco`, `mujoco_py`, `pybullet` (importlib, no installation),
  2. EMITS a model description from this build's OWN tables -- MJCF text from earl_benchmark_amd/models/<name>.npz (bodies, joints, primitive geoms,
     actuators, equalities: the numbers tools/mjcf_compile.py extracted, never the reference's files), URDF text from models/minitaur_links.npz --,
  3. loads it into whatever simulator was found, puts both sides into the same state, advances ONE env step (frame_skip timesteps) with the same
     control, and records the differences in profiles/<tag>_simulator_pin.json.

Without a simulator it writes the probe result and stops (exit code 0): that file is then the evidence that the pin could not be taken on this host.
The emitters are unit-tested without any simulator (tests/test_pin_tool.py: emitted MJCF -> this build's own MJCF compiler -> the same tables).

    python tools/pin_with_simulator.py [--tag r04] [--models sawyer_door,sawyer_peg,kitchen,minitaur] [--emit-only DIR]
"""
import argparse
import importlib.util
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL_DIR = os.path.join(ROOT, 'earl_benchmark_amd', 'models')
GEOM_TYPES = {0: 'plane', 1: 'sphere', 2: 'capsule', 3: 'cylinder', 4: 'box', 5: 'mesh'}
# <option> facts the tables do not carry: the Sawyer scene asks for Newton / 50 iterations / 1e-10 / elliptic cones
# (metaworld_assets/scene/basic_scene.xml:2), the kitchen keeps MuJoCo's defaults (third_party/franka/assets/assets.xml:7 sets the timestep only)
OPTIONS = {'sawyer_door': dict(solver='Newton', iterations='50', tolerance='1e-10', cone='elliptic', jacobian='dense'),
           'sawyer_peg': dict(solver='Newton', iterations='50', tolerance='1e-10', cone='elliptic', jacobian='dense'),
           'kitchen': {}}
FRAME_SKIP = {'sawyer_door': 5, 'sawyer_peg': 5, 'kitchen': 40, 'minitaur': 5}


def fmt(v):
  return ' '.join(repr(float(x)) for x in np.atleast_1d(v))


def load_tables(name):
  with np.load(os.path.join(MODEL_DIR, name + '.npz'), allow_pickle=False) as z:
    return {k: z[k] for k in z.files}


def emit_mjcf(t, name=None, cone=None):
  """MJCF text of one of this build's full-body tables (models/<name>.npz).  Every body gets an explicit <inertial> (the tables hold compiled masses and
  inertias: nothing is left to inertiafromgeom); mesh geoms (no mesh data ships) are left out -- they only ever carried mass, which is in the <inertial>s.
  Joint, geom and body ORDER is the tables' order, so qpos / qvel / ctrl index alike on both sides."""
  name = name or str(t['name'])
  root = ET.Element('mujoco', model=f'{name} (emitted by tools/pin_with_simulator.py from this build\'s own tables)')
  ET.SubElement(root, 'compiler', angle='radian', inertiafromgeom='false', inertiagrouprange='5 5', autolimits='false')    # (no geom is in group 5: masses come from the <inertial>s only)
  opt = dict(OPTIONS.get(name, {}))
  if cone:
    opt['cone'] = cone
  ET.SubElement(root, 'option', timestep=repr(float(t['timestep'])), gravity=fmt(t['gravity']), **opt)
  ET.SubElement(root, 'size', njmax='2000', nconmax='500')
  world = ET.SubElement(root, 'worldbody')
  nb = len(t['body_parent'])
  els = {0: world}
  bnames = [str(x) or f'body{i}' for i, x in enumerate(t['body_names'])]
  jnames = [str(x) or f'joint{i}' for i, x in enumerate(t['joint_names'])]
  jnt_of_body = {}
  for ji, b in enumerate(t['jnt_body']):
    jnt_of_body.setdefault(int(b), []).append(ji)
  for b in range(1, nb):
    free = bool(t['body_free'][b]) if 'body_free' in t else False
    pos, quat = (t['body_qpos0'][b][:3], t['body_qpos0'][b][3:]) if free else (t['body_pos'][b], t['body_quat'][b])
    attrs = dict(name=bnames[b], pos=fmt(pos), quat=fmt(quat))
    if int(t['body_mocap'][b]):
      attrs['mocap'] = 'true'
    el = ET.SubElement(els[int(t['body_parent'][b])], 'body', **attrs)
    els[b] = el
    if float(t['body_mass'][b]) > 0 or np.any(t['body_ipos'][b] != 0) or np.any(t['body_iquat'][b] != np.array([1.0, 0, 0, 0])):      # (massless <inertial> frames exist in these models)
      ET.SubElement(el, 'inertial', pos=fmt(t['body_ipos'][b]), quat=fmt(t['body_iquat'][b]), mass=repr(float(t['body_mass'][b])),
                    diaginertia=fmt(t['body_inertia'][b]))
    js = jnt_of_body.get(b, [])
    if free:
      assert len(js) == 6
      ET.SubElement(el, 'joint', name=jnames[js[0]][:-3], type='free', damping=repr(float(t['jnt_damping'][js[0]])), armature=repr(float(t['jnt_armature'][js[0]])))
    else:
      for ji in js:
        a = dict(name=jnames[ji], type='hinge' if int(t['jnt_type'][ji]) == 0 else 'slide', pos=fmt(t['jnt_pos'][ji]), axis=fmt(t['jnt_axis'][ji]),
                 limited='true' if int(t['jnt_limited'][ji]) else 'false', range=fmt(t['jnt_range'][ji]), damping=repr(float(t['jnt_damping'][ji])),
                 armature=repr(float(t['jnt_armature'][ji])), solreflimit=fmt(t['jnt_solref'][ji]), solimplimit=fmt(t['jnt_solimp'][ji]))
        for key, attr in (('jnt_frictionloss', 'frictionloss'), ('jnt_stiffness', 'stiffness'), ('jnt_springref', 'springref')):
          if key in t and float(t[key][ji]) != 0.0:
            a[attr] = repr(float(t[key][ji]))
        ET.SubElement(el, 'joint', **a)
  for g in range(len(t['geom_body'])):
    gt = GEOM_TYPES[int(t['geom_type'][g])]
    if gt == 'mesh':
      continue
    size = t['geom_size'][g]
    nsz = {'plane': 3, 'sphere': 1, 'capsule': 2, 'cylinder': 2, 'box': 3}[gt]
    nsz = max([nsz] + [k + 1 for k in range(3) if size[k] != 0])       # (these files give some shapes more size numbers than the shape uses: kept, MuJoCo ignores them)
    a = dict(type=gt, size=fmt(size[:nsz]), pos=fmt(t['geom_pos'][g]), quat=fmt(t['geom_quat'][g]), contype=str(int(t['geom_contype'][g])),
             conaffinity=str(int(t['geom_conaffinity'][g])), condim=str(int(t['geom_condim'][g])), friction=fmt(t['geom_friction'][g]),
             solref=fmt(t['geom_solref'][g]), solimp=fmt(t['geom_solimp'][g]), margin=repr(float(t['geom_margin'][g])))
    if str(t['geom_names'][g]):
      a['name'] = str(t['geom_names'][g])
    ET.SubElement(els[int(t['geom_body'][g])], 'geom', **a)
  for s in range(len(t['site_body'])):
    a = dict(pos=fmt(t['site_pos'][s]))
    if str(t['site_names'][s]):
      a['name'] = str(t['site_names'][s])
    ET.SubElement(els[int(t['site_body'][s])], 'site', **a)
  # element order = table order: the tables number geoms / sites in document order, where world-level geoms and sites stand BETWEEN top-level bodies
  top = {}
  for b in range(1, nb):
    r = b
    while int(t['body_parent'][r]) != 0:
      r = int(t['body_parent'][r])
    top[b] = r
  for kind, owner in (('geom', t['geom_body']), ('site', t['site_body'])):
    emitted = [i for i in range(len(owner)) if kind == 'site' or GEOM_TYPES[int(t['geom_type'][i])] != 'mesh']
    first_after = {}                                     # top-level body -> smallest index of this kind in its subtree
    for i in emitted:
      if int(owner[i]) != 0:
        first_after.setdefault(top[int(owner[i])], i)
    world_items = [ch for ch in list(world) if ch.tag == kind]
    world_idx = [i for i in emitted if int(owner[i]) == 0]
    for ch, i in zip(world_items, world_idx):
      world.remove(ch)
      later = [els[b] for b in sorted(first_after) if first_after[b] > i]
      if later:
        world.insert(list(world).index(later[0]), ch)
      else:
        world.append(ch)
  if len(t['act_joint']):
    act = ET.SubElement(root, 'actuator')
    for k, ji in enumerate(t['act_joint']):
      a = dict(joint=jnames[int(ji)], kp=repr(float(t['act_kp'][k])), ctrllimited='true', ctrlrange=fmt(t['act_ctrlrange'][k]))
      if 'act_forcerange' in t and np.all(np.isfinite(t['act_forcerange'][k])):
        a.update(forcelimited='true', forcerange=fmt(t['act_forcerange'][k]))
      ET.SubElement(act, 'position', **a)
  if len(t['weld_body1']) or len(t.get('jeq_joint1', ())):
    eq = ET.SubElement(root, 'equality')
    for k in range(len(t['weld_body1'])):
      ET.SubElement(eq, 'weld', body1=bnames[int(t['weld_body1'][k])], body2=bnames[int(t['weld_body2'][k])], solref=fmt(t['weld_solref'][k]),
                    solimp=fmt(t['weld_solimp'][k]))
    for k in range(len(t.get('jeq_joint1', ()))):
      ET.SubElement(eq, 'joint', joint1=jnames[int(t['jeq_joint1'][k])], joint2=jnames[int(t['jeq_joint2'][k])],
                    polycoef=fmt([t['jeq_coef'][k][0], t['jeq_coef'][k][1], 0, 0, 0]), solref=fmt(t['jeq_solref'][k]), solimp=fmt(t['jeq_solimp'][k]))
  if 'key_qpos' in t and len(t['key_qpos']):
    kf = ET.SubElement(root, 'keyframe')
    ET.SubElement(kf, 'key', qpos=fmt(t['key_qpos']))
  ET.indent(root, space=' ')
  return ET.tostring(root, encoding='unicode')


def quat_to_rpy(q):
  w, x, y, z = [float(v) for v in q]
  r = np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y))
  p = np.arcsin(np.clip(2 * (w * y - z * x), -1, 1))
  yw = np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))
  return r, p, yw


def emit_urdf_minitaur(t):
  """URDF text of this build's own minitaur (models/minitaur_links.npz; the reference's pybullet_data/quadruped/minitaur.urdf is not in its tree): the root
  body (link table rows 0-5 are its six dofs) becomes the floating base link, each hinge row a continuous joint + link; collision = the model's spheres.
  The four loop closures are NOT URDF: the pin adds them with createConstraint(JOINT_POINT2POINT), as the reference does (envs/minitaur.py:212-217)."""
  base = int(t['ball_dof']) + 2
  names = [str(x) for x in t['link_names']]
  robot = ET.Element('robot', name='minitaur (this build\'s own authoring, emitted from models/minitaur_links.npz)')

  def link(l):
    el = ET.SubElement(robot, 'link', name=names[l])
    ine = ET.SubElement(el, 'inertial')
    ET.SubElement(ine, 'origin', xyz=fmt(t['com'][l]), rpy='0 0 0')
    ET.SubElement(ine, 'mass', value=repr(float(t['mass'][l])))
    I = t['inertia'][l]
    ET.SubElement(ine, 'inertia', ixx=repr(float(I[0])), iyy=repr(float(I[1])), izz=repr(float(I[2])), ixy=repr(float(I[3])), ixz=repr(float(I[4])), iyz=repr(float(I[5])))
    for si in np.nonzero(t['col_sph_link'] == l)[0]:
      c = ET.SubElement(el, 'collision')
      ET.SubElement(c, 'origin', xyz=fmt(t['col_sph_pos'][si]), rpy='0 0 0')
      ET.SubElement(ET.SubElement(c, 'geometry'), 'sphere', radius=repr(float(t['col_sph_r'][si])))
  link(base)
  for l in range(base + 1, len(t['parent'])):
    link(l)
    jt = ET.SubElement(robot, 'joint', name=names[l] + '_hinge', type='continuous')
    ET.SubElement(jt, 'parent', link=names[int(t['parent'][l])])
    ET.SubElement(jt, 'child', link=names[l])
    ET.SubElement(jt, 'origin', xyz=fmt(t['tpos'][l]), rpy=fmt(quat_to_rpy(t['tquat'][l])))
    ET.SubElement(jt, 'axis', xyz=fmt(t['jaxis'][l]))
    ET.SubElement(jt, 'dynamics', damping=repr(float(t['jnt_damping'][l])), friction='0')
  ET.indent(robot, space=' ')
  return ET.tostring(robot, encoding='unicode')


def probe():
  found = {}
  for name in ('mujoco', 'mujoco_py', 'pybullet'):
    try:
      found[name] = importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
      found[name] = False
  return found


def pin_mujoco(name, xml, tables):
  """one env step of the emitted model in `mujoco` (the maintained bindings) next to the HIP stepper, from the model's key / zero pose with the fingers
  commanded half open and the mocap target moved 1 cm: qpos / qvel differences after frame_skip timesteps"""
  import mujoco
  import torch
  sys.path.insert(0, ROOT)
  from earl_benchmark_amd import physics
  m = mujoco.MjModel.from_xml_string(xml)
  d = mujoco.MjData(m)
  if m.nkey:
    mujoco.mj_resetDataKeyframe(m, d, 0)
  mujoco.mj_forward(m, d)
  stepper = physics.Stepper(name)
  nv, nq = m.nv, m.nq
  q0, v0 = d.qpos.copy(), d.qvel.copy()
  mocap = d.mocap_pos[0].copy() + np.array([0.01, 0.0, 0.0])
  d.mocap_pos[0] = mocap
  ctrl = np.array([0.5 * (lo + hi) for lo, hi in m.actuator_ctrlrange])
  d.ctrl[:] = ctrl
  for _ in range(FRAME_SKIP[name]):
    mujoco.mj_step(m, d)
  tq = torch.tensor(q0[None], dtype=torch.float64, device='cuda')
  tv = torch.tensor(v0[None], dtype=torch.float64, device='cuda')
  stepper.step(tq, tv, torch.tensor(mocap[None], device='cuda'), torch.tensor(d.mocap_quat[:1].copy(), device='cuda'),
               torch.tensor(ctrl[None], device='cuda'), nsub=FRAME_SKIP[name])
  dq, dv = tq.cpu().numpy()[0] - d.qpos, tv.cpu().numpy()[0] - d.qvel
  return {'model': name, 'simulator': f'mujoco {mujoco.__version__}', 'nq': int(nq), 'nv': int(nv), 'timesteps': FRAME_SKIP[name],
          'max_abs_dqpos': float(np.abs(dq).max()), 'max_abs_dqvel': float(np.abs(dv).max()), 'dqpos': dq.tolist(), 'dqvel': dv.tolist()}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--tag', default='r04')
  ap.add_argument('--models', default='sawyer_door,sawyer_peg,kitchen,minitaur')
  ap.add_argument('--emit-only', default=None, help='write the emitted MJCF / URDF texts into this directory and stop')
  a = ap.parse_args()
  models = [x for x in a.models.split(',') if x]
  texts = {}
  for name in models:
    texts[name] = emit_urdf_minitaur(load_tables('minitaur_links')) if name == 'minitaur' else emit_mjcf(load_tables(name), name)
  if a.emit_only:
    os.makedirs(a.emit_only, exist_ok=True)
    for name, txt in texts.items():
      with open(os.path.join(a.emit_only, name + ('.urdf' if name == 'minitaur' else '.xml')), 'w') as f:
        f.write(txt)
    return 0
  found = probe()
  out = {'probed': sorted(found), 'found': [k for k, v in found.items() if v], 'emitted': {k: {'bytes': len(v), 'kind': 'urdf' if k == 'minitaur' else 'mjcf'} for k, v in texts.items()},
         'pins': [], 'status': None}
  if found.get('mujoco'):
    for name in models:
      if name != 'minitaur':
        try:
          out['pins'].append(pin_mujoco(name, texts[name], load_tables(name)))
        except Exception as e:      # noqa: BLE001  (a failed pin is a result too)
          out['pins'].append({'model': name, 'simulator': 'mujoco', 'error': repr(e)[:500]})
    out['status'] = 'pinned against mujoco (see pins)'
  elif found.get('mujoco_py') or found.get('pybullet'):
    out['status'] = ('found ' + ', '.join(out['found']) + ' but not the `mujoco` bindings this tool drives; the emitted model texts load in either '
                     '(mujoco_py.load_model_from_xml / pybullet.loadURDF): extend pin_mujoco accordingly')
  else:
    out['status'] = 'no simulator importable on this host: the rigid-body rows stay parity-unpinned (DESIGN.md); the emitters are unit-tested in tests/test_pin_tool.py'
  path = os.path.join(ROOT, 'profiles', f'{a.tag}_simulator_pin.json')
  with open(path, 'w') as f:
    json.dump(out, f, indent=1)
  print(json.dumps({k: out[k] for k in ('found', 'status')}))
  return 0


if __name__ == '__main__':
  sys.exit(main())
