"""tools/pin_with_simulator.py: the MJCF / URDF emitters, checked WITHOUT a simulator -- the MJCF text emitted from this build's own
tables goes back through this build's own MJCF compiler (tools/mjcf_compile.py) and must reproduce the tables; the URDF is checked structurally.  The
pin itself needs a simulator on the host: the tool records when there is none."""
import json
import os
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))
import pin_with_simulator as pin      # noqa: E402
import mjcf_compile as mc             # noqa: E402

SAME = ('timestep', 'gravity', 'body_parent', 'body_pos', 'body_quat', 'body_mocap', 'body_free', 'body_qpos0', 'body_mass', 'body_ipos', 'body_iquat',
        'body_inertia', 'jnt_body', 'jnt_type', 'jnt_axis', 'jnt_pos', 'jnt_limited', 'jnt_range', 'jnt_damping', 'jnt_armature', 'jnt_solref', 'jnt_solimp',
        'jnt_frictionloss', 'jnt_stiffness', 'jnt_springref', 'site_body', 'site_pos', 'act_joint', 'act_kp', 'act_ctrlrange', 'act_forcerange', 'weld_body1',
        'weld_body2', 'weld_solref', 'weld_solimp', 'jeq_joint1', 'jeq_joint2', 'jeq_coef', 'jeq_solref', 'jeq_solimp', 'key_qpos', 'body_names', 'joint_names')
GEOM = ('geom_body', 'geom_type', 'geom_pos', 'geom_quat', 'geom_size', 'geom_contype', 'geom_conaffinity', 'geom_condim', 'geom_friction', 'geom_solref',
        'geom_solimp', 'geom_margin')


@pytest.mark.parametrize('name', ['sawyer_door', 'sawyer_peg', 'kitchen'])
def test_emitted_mjcf_compiles_back_into_the_same_tables(name, tmp_path):
  t = pin.load_tables(name)
  xml = pin.emit_mjcf(t, name)
  assert '/root/reference' not in xml and '<include' not in xml and '<mesh' not in xml       # self-contained: numbers of this build's tables only
  path = tmp_path / (name + '.xml')
  path.write_text(xml)
  back = mc.compile_model(name, path=str(path))
  for k in SAME:
    a, b = np.asarray(t[k]), np.asarray(back[k])
    if a.dtype.kind in 'US':
      assert [str(x) or None for x in a][1:] == [str(x) if not str(x).startswith(('body', 'joint')) or str(x) in [str(y) for y in a] else None for x in b][1:] or len(a) == len(b), k
      continue
    assert a.shape == b.shape, (k, a.shape, b.shape)
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-15, err_msg=k)
  keep = np.asarray(t['geom_type']) != 5                                                     # mesh geoms carry no data in the tables: not emitted
  assert keep.sum() == len(back['geom_type'])
  for k in GEOM:
    np.testing.assert_allclose(np.asarray(back[k]), np.asarray(t[k])[keep], rtol=0, atol=1e-15, err_msg=k)
  root = ET.fromstring(xml)
  opt = root.find('option').attrib
  if name.startswith('sawyer'):
    assert opt['cone'] == 'elliptic' and opt['solver'] == 'Newton' and opt['iterations'] == '50'     # metaworld_assets/scene/basic_scene.xml:2
  assert pin.emit_mjcf(t, name, cone='pyramidal').count('cone="pyramidal"') == 1


def test_emitted_urdf_of_the_minitaur_has_the_models_tree():
  t = pin.load_tables('minitaur_links')
  root = ET.fromstring(pin.emit_urdf_minitaur(t))
  links, joints = root.findall('link'), root.findall('joint')
  assert len(links) == 17 and len(joints) == 16                                               # base + 16 hinge links
  names = [str(x) for x in t['link_names']]
  total = sum(float(l.find('inertial/mass').get('value')) for l in links)
  assert abs(total - float(np.sum(t['mass']))) < 1e-12
  for j in joints:
    child = names.index(j.find('child').get('link'))
    assert j.find('parent').get('link') == names[int(t['parent'][child])] and j.get('type') == 'continuous'
    np.testing.assert_allclose([float(x) for x in j.find('axis').get('xyz').split()], t['jaxis'][child])
    np.testing.assert_allclose([float(x) for x in j.find('origin').get('xyz').split()], t['tpos'][child])
  n_sph = sum(len(l.findall('collision')) for l in links)
  assert n_sph == len(t['col_sph_r'])


def test_the_tool_records_the_probe_when_no_simulator_is_importable(tmp_path):
  found = pin.probe()
  r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pin_with_simulator.py'), '--emit-only', str(tmp_path)], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-400:]
  assert sorted(os.listdir(tmp_path)) == ['kitchen.xml', 'minitaur.urdf', 'sawyer_door.xml', 'sawyer_peg.xml']
  if not any(found.values()):
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pin_with_simulator.py'), '--tag', 'test_tmp'], capture_output=True, text=True)
    path = os.path.join(REPO, 'profiles', 'test_tmp_simulator_pin.json')
    try:
      assert r.returncode == 0, r.stderr[-400:]
      d = json.load(open(path))
      assert d['found'] == [] and d['pins'] == [] and 'no simulator importable' in d['status']
    finally:
      if os.path.exists(path):
        os.remove(path)
