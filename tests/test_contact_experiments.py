"""Round 5, VERDICT r04 item 1: the door's gripper-handle contacts as the reference's simulator models them -- (a) claw plate and pad as two boxes, (b) the handle's four
cylinders as cylinders with ONE contact per geom pair from a box-cylinder narrow phase (Minkowski portal refinement, the routine MuJoCo sends this pair to), (c) the
condim-4 torsional row -- exist in the C restatement (oracle/physics_oracle.c) and in experimental collision tables (tools/mjcf_compile.py DOOR_CONTACTS=...), and were
ablated on the ten recorded door episodes (tools/door_contact_ablation.py -> profiles/r05_door_contact_ablation.json).  They are NOT in the shipped tables or kernels: every
variant loses all five reverse episodes.  These tests pin the narrow phase's known answers and the ablation's numbers."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle.tabletop_oracle import lib

# a finger plate as the door model has it: long axis pointing down (world -z), thin along y, 30 mm wide along x
RB = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], float)        # columns = the box axes in the world
H = np.array([0.045, 0.005, 0.015])
CYL = dict(c=np.array([0.0, 0.0, 0.1]), a=np.array([1.0, 0.0, 0.0]), hl=0.054, r=0.023)
MARGIN = 0.001


def mpr(pb, Rb=RB, h=H, margin=MARGIN, **cyl):
  cy = dict(CYL, **cyl)
  out = np.zeros(7)
  f = lib().oracle_mpr_box_cylinder
  f.restype = C.c_int
  arr = lambda x: np.ascontiguousarray(x, np.float64).ctypes.data_as(C.c_void_p)
  hit = f(arr(pb), arr(np.asarray(Rb).ravel()), arr(h), arr(cy['c']), arr(cy['a']), C.c_double(cy['hl']), C.c_double(cy['r']), C.c_double(margin), out.ctypes.data_as(C.c_void_p))
  return hit, out[0], out[1:4], out[4:7]


@pytest.mark.parametrize('gap', [0.0005, 0.0, -0.0004, -0.002])
def test_plate_face_against_the_cylinders_side(gap):
  """the plate's flat face (normal -y) against the rod's rear-most line, well inside the face: distance = the gap (negative: penetration), normal = the face normal
  pointing from the box to the cylinder, position midway between the two surfaces at the rod's equator"""
  pb = CYL['c'] + np.array([0.004, CYL['r'] + H[1] + gap, 0.02])
  hit, dist, n, pos = mpr(pb)
  assert hit == 1
  np.testing.assert_allclose(dist, gap, atol=1e-9)
  np.testing.assert_allclose(n, [0, -1, 0], atol=1e-7)
  np.testing.assert_allclose(pos[1:], [CYL['r'] + 0.5 * gap, 0.1], atol=1e-3)      # (a barycentric mix of support points of the two inflated shapes: within the margin)


def test_beyond_the_margin_there_is_no_contact_and_a_cap_contact_has_the_axis_as_normal():
  pb = CYL['c'] + np.array([0.004, CYL['r'] + H[1] + MARGIN + 1e-5, 0.02])
  assert mpr(pb)[0] == 0
  # the plate's face against the cylinder's flat END: turn the cylinder so that its axis is the plate's thin direction
  a = np.array([0.0, 1.0, 0.0])
  pb = CYL['c'] + np.array([0.002, CYL['hl'] + H[1] + 0.0003, 0.01])
  hit, dist, n, pos = mpr(pb, a=a)
  assert hit == 1
  np.testing.assert_allclose(dist, 0.0003, atol=1e-9)
  np.testing.assert_allclose(n, [0, -1, 0], atol=1e-7)


def test_at_the_plates_tip_the_portal_direction_is_the_centre_ray_not_the_surface_normal():
  """what the ablation turned on: with the plate's bottom edge level with the rod's axis (within a millimetre) the origin ray leaves the Minkowski difference through its
  rounded corner, the final portal is a sliver along the rod, and the reported direction is the ray from the plate's centre to the rod's centre -- 60 degrees off the
  surface normal (0, -1, 0) at the touching point.  Five millimetres lower the same routine returns the face normal."""
  tip_level = CYL['c'] + np.array([0.005, 0.028, 0.0456])        # box centre 45.6 mm above the axis: bottom edge 0.6 mm below it
  hit, dist, n, pos = mpr(tip_level)
  assert hit == 1 and n[2] < -0.8 and abs(n[1]) < 0.6
  ray = -(tip_level - CYL['c']); ray[0] = 0; ray /= np.linalg.norm(ray)
  np.testing.assert_allclose(n[1:], ray[1:], atol=2e-2)
  lower = CYL['c'] + np.array([0.005, 0.0275, 0.040])
  np.testing.assert_allclose(mpr(lower)[2], [0, -1, 0], atol=1e-7)


def test_the_ablation_numbers_are_what_profiles_holds():
  """the committed table: the shipped set ('chains') against the three experimental ones, all ten recorded door episodes, open loop, constants frozen"""
  sys.path.insert(0, os.path.join(REPO, 'tools'))
  import door_contact_ablation as dca
  from oracle import physics_c
  physics_c.set_threads(min(8, os.cpu_count() or 1))
  want = json.load(open(os.path.join(REPO, 'profiles', 'r05_door_contact_ablation.json')))
  for v in ('chains', 'cyl+split+tor'):
    got = dca.run(v)
    for d in ('forward', 'reverse'):
      np.testing.assert_allclose(got[d]['obj_rms_mm'], want[v][d]['obj_rms_mm'], atol=0.35)
      assert got[d]['reached'] == want[v][d]['reached']
  assert want['chains']['reverse']['reached'] == 1 and all(want[v]['reverse']['reached'] == 0 for v in ('cyl', 'cyl+split', 'cyl+split+tor'))


def test_the_product_refuses_the_experimental_tables():
  """the kernels implement neither cylinder pairs nor torsional rows: the device-side loader must refuse such tables rather than read a cylinder as a sphere (no GPU needed:
  the check precedes every device call)"""
  from earl_benchmark_amd import _abi, physics
  _, tables = physics.load_link_model('sawyer_door_cyl_split_tor')
  col = physics.load_collision_model(tables)
  assert any(col.pair_kind[i] == 2 for i in range(col.n_pair)) and any(col.cls_mu_tor[:])
  with pytest.raises(_abi.EarlHipError, match='cylinder pairs'):
    physics.DeviceModel('sawyer_door_cyl_split_tor', device='cpu')
  _, shipped = physics.load_link_model('sawyer_door')
  c0 = physics.load_collision_model(shipped)
  assert not any(c0.pair_kind[i] == 2 for i in range(c0.n_pair)) and not any(c0.cls_mu_tor[:])
