"""Kitchen model tables and CPU statement (no GPU): what the compiled tables must say about the reference's MJCF, and first-principles checks
of the features the kitchen adds to the stepper (joint couplings, dry friction, springs, force-limited actuators).  Parity with MuJoCo is
UNPINNED for this env (no recordings, no simulator) -- these tests pin the provenance of the numbers and the internal consistency."""
import os

import numpy as np
import pytest

from conftest import REPO, load_golden

LINKS = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'kitchen_links.npz')
FULL = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'kitchen.npz')
INIT_QPOS = None


@pytest.fixture(scope='module')
def lm():
  from oracle import physics_oracle as po
  return po.LinkModel(LINKS)


def test_model_facts_match_the_mjcf_and_the_robot_config():
  """42 bodies + world, nq = nv = 23 (9 robot + 14 fixtures), 229 geoms, nu = 2, one weld + five joint couplings,
  dt = 0.002; joint ranges = the pos bounds the reference reads from franka_config.xml (golden recorded from the reference)"""
  z = np.load(FULL)
  assert len(z['body_parent']) == 43 and len(z['jnt_body']) == 23 and len(z['geom_body']) == 229 and len(z['act_joint']) == 2
  assert float(z['timestep']) == 0.002 and len(z['weld_body1']) == 1 and len(z['jeq_joint1']) == 5
  names = [str(x) for x in z['joint_names']]
  assert names[:9] == [f'panda0_joint{k}' for k in range(1, 8)] + ['panda0_finger_joint1', 'panda0_finger_joint2']
  assert names[9:] == ['knob_Joint_1', 'burner_Joint_1', 'knob_Joint_2', 'burner_Joint_2', 'knob_Joint_3', 'burner_Joint_3', 'knob_Joint_4', 'burner_Joint_4',
                       'lightswitch_joint', 'light_joint', 'slidedoor_joint', 'leftdoorhinge', 'rightdoorhinge', 'microjoint']
  # joint limits of the simulator model (third_party/franka/assets/chain0.xml:10-42); the robot CONFIG's position bounds (franka_config.xml,
  # in the kitchen_step golden) are a different, coarser table used only by the action / reset clipping of the glue
  np.testing.assert_allclose(z['jnt_range'][:7], [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.4], [-2.8973, 2.8973],
                                                  [-1.6573, 2.1127], [-2.8973, 2.8973]])
  np.testing.assert_allclose(z['jnt_range'][7:9], [[0, 0.04], [0, 0.04]])
  assert load_golden('kitchen_step')['kitchen_pos_bound'].shape == (23, 2)
  np.testing.assert_allclose(z['jeq_coef'], [[0, 174]] * 4 + [[0, 14]])
  assert list(z['jeq_joint1']) == [9, 11, 13, 15, 17] and list(z['jeq_joint2']) == [10, 12, 14, 16, 18]
  np.testing.assert_allclose(z['jnt_frictionloss'][17:], [1, 1, 2, 2, 2, 2]); assert not z['jnt_frictionloss'][:17].any()
  np.testing.assert_allclose(z['jnt_damping'][:9], [100] * 4 + [10] * 3 + [100] * 2)
  np.testing.assert_allclose(z['act_forcerange'], [[-70, 70]] * 2); np.testing.assert_allclose(z['act_kp'], [500, 500])
  np.testing.assert_allclose(z['weld_solimp'][0], [0.4, 0.85, 0.1, 0.5, 2.0])
  mass = {str(n): float(m) for n, m in zip(z['body_names'], z['body_mass'])}
  for k, m in zip(range(8), (2.91242, 2.7063, 2.73046, 2.04104, 2.08129, 3.00049, 1.3235, 0.2 + 0.81909)):     # chain0.xml: masses on the collision hulls
    assert abs(mass[f'panda0_link{k}'] - m) < 1e-9
  # the initial / goal tables the loader serves are reachable states of this model
  init, goal = load_golden('loader_tables')['kitchen_initial_states'], load_golden('loader_tables')['kitchen_goal_states']
  assert init.shape == (6, 23) and goal.shape == (1, 23)
  lo, hi = z['jnt_range'][9:, 0] - 0.02, z['jnt_range'][9:, 1] + 0.02
  assert ((init[:, 9:] >= lo) & (init[:, 9:] <= hi)).all() and ((goal[:, 9:] >= lo) & (goal[:, 9:] <= hi)).all()


def test_mesh_inertia_of_a_known_solid(tmp_path):
  """the mesh mass-property routine on a closed box mesh: mass, centre and inertia tensor of a uniform box"""
  import struct
  import sys
  sys.path.insert(0, os.path.join(REPO, 'tools'))
  import mjcf_compile as mc
  h = np.array([0.3, 0.2, 0.1]); c = np.array([0.05, -0.02, 0.4])
  v = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float) * h + c
  quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
  tris = [(q[0], q[1], q[2]) for q in quads] + [(q[0], q[2], q[3]) for q in quads]
  p = tmp_path / 'box.stl'
  with open(p, 'wb') as f:
    f.write(b'\0' * 80 + struct.pack('<I', len(tris)))
    for t in tris:
      f.write(struct.pack('<12fH', 0, 0, 0, *v[t[0]], *v[t[1]], *v[t[2]], 0))
  com, I = mc.mesh_inertia(str(p), [1, 1, 1], 2.5)
  np.testing.assert_allclose(com, c, atol=1e-6)        # (the STL stores float32)
  np.testing.assert_allclose(I, np.diag(2.5 / 3 * np.array([h[1]**2 + h[2]**2, h[0]**2 + h[2]**2, h[0]**2 + h[1]**2])), atol=1e-6)


def test_weld_pulls_the_arm_and_the_fixtures_stay_put(lm):
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  q, v = INIT_QPOS.copy(), np.zeros(23)
  mp, mq = np.array(MIDPOINT_POS), np.array(lm.weld_mocap_quat)
  pos, quat, _ = lm.kinematics(q)
  d0 = np.linalg.norm(lm.attachment(pos, quat, 0)[0] - mp)
  for _ in range(120):
    q, v, out = lm.step(q, v, np.array([0.04, 0.0]), mp, mq)
  d1 = np.linalg.norm(lm.attachment(out['pos'], out['quat'], 0)[0] - mp)
  assert d1 < 0.3 * d0 and np.isfinite(q).all()                   # the welded link moves to the mocap (slowly: joint damping 100)
  assert np.abs(q[9:] - np.clip(INIT_QPOS[9:], lm.jnt_range[9:, 0], lm.jnt_range[9:, 1])).max() < 5e-3   # nothing touches the fixtures
  assert abs(q[7] - 0.04) < 5e-3 and q[8] < 0.03                  # finger actuators track their (clamped) targets


def test_joint_coupling_dry_friction_spring_and_force_limit(lm):
  """first principles on single fixtures (each is its own tree, so its row of M is a scalar):
  * knob / burner coupling: a turned knob pulls its burner towards q_knob = 174 q_burner (soft equality);
  * dry friction holds a door against a small torque and lets it slide under a large one, with force exactly +-frictionloss when sliding;
  * a joint spring pulls towards springref; a force-limited actuator saturates at its forcerange"""
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  mp, mq = np.array(MIDPOINT_POS), np.array(lm.weld_mocap_quat)
  base = np.clip(INIT_QPOS, None, None).copy(); base[9:] = 0.0
  # coupling
  q = base.copy(); q[9] = -0.87                                      # knob 1 turned, burner 1 at rest
  r = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)
  assert r['qacc'][10] < -1.0 and r['qacc'][9] > 0                   # burner accelerates towards -0.87 / 174, knob is pulled back
  qq, vv = q.copy(), np.zeros(23)
  for _ in range(400):
    qq, vv, _ = lm.step(qq, vv, np.array([0.04, 0.0]), mp, mq)
  assert abs(qq[9] - 174 * qq[10]) < 0.05 and -0.009 - 1e-3 < qq[10] < 0
  # dry friction: microwave door (dof 22, frictionloss 2, damping 2, no gravity torque about its vertical hinge)
  M22 = lm.forward(base, np.zeros(23), np.array([0.04, 0.0]), mp, mq)['M'][22, 22]
  for torque, slides in ((1.0, False), (5.0, True)):
    lm2 = lm
    q = base.copy(); q[22] = -0.5
    v = np.zeros(23)
    # apply the torque through gravity-free means: a velocity kick equivalent over one step is awkward; use the spring table instead
    k_save, ref_save = lm.jnt_stiffness.copy(), lm.jnt_springref.copy()
    try:
      lm.jnt_stiffness[22] = 1.0; lm.jnt_springref[22] = q[22] + torque      # spring torque = +torque at this angle
      r = lm.forward(q, v, np.array([0.04, 0.0]), mp, mq)
    finally:
      lm.jnt_stiffness[:], lm.jnt_springref[:] = k_save, ref_save
    if slides:
      np.testing.assert_allclose(r['qacc'][22], (torque - 2.0) / M22, rtol=1e-6)      # friction saturated at frictionloss = 2
    else:
      # held, softly: in the quadratic zone the row is a regulariser R = (1 - d) / d / M = M^-1 / 9 (d = 0.9 at zero residual), so a = torque / (10 M)
        np.testing.assert_allclose(r['qacc'][22], torque / (10 * M22), rtol=1e-6)
  # spring on burner 2 (stiffness 1, springref 0): force -k q
  q = base.copy(); q[12] = -0.005; q[11] = 174 * -0.005
  r0 = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)
  k_save = lm.jnt_stiffness.copy()
  try:
    lm.jnt_stiffness[12] = 0.0
    r1 = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)
  finally:
    lm.jnt_stiffness[:] = k_save
  assert r0['qacc'][12] > r1['qacc'][12]                              # the spring pushes the burner back up
  # force limit: finger far from its target -> kp * error = 500 * 0.04 = 20 N < 70: not saturated; scale kp to saturate
  kp_save = lm.act_kp.copy()
  try:
    q = base.copy(); q[7] = 0.0
    a_lo = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)['qacc'][7]
    lm.act_kp[:] = 50000.0                                           # 2000 N demanded, 70 N delivered
    a_hi = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)['qacc'][7]
    lm.act_kp[:] = 0.0
    a_0 = lm.forward(q, np.zeros(23), np.array([0.04, 0.0]), mp, mq)['qacc'][7]
  finally:
    lm.act_kp[:] = kp_save
  np.testing.assert_allclose((a_hi - a_0) / (a_lo - a_0), 70.0 / 20.0, rtol=1e-3)     # affine in the actuator force (the other finger, 18.6 N -> 70 N, couples in at 1e-4)


def test_kitchen_oracle_env_step_runs(lm):
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  from oracle import glue_oracle as go
  from oracle.kitchen_oracle import KitchenOracle
  g = load_golden('kitchen_step')
  p = go.kitchen_params(g['kitchen_pos_bound'], g['kitchen_vel_bound'], g['kitchen_pos_noise_amp'])
  env = KitchenOracle(p, lm)
  goal = load_golden('loader_tables')['kitchen_goal_states'][0]
  env.set(INIT_QPOS, np.zeros(23), MIDPOINT_POS, goal, INIT_QPOS[:9])
  obs, r, s, _ = env.step(np.array([0.5, -0.5, 0.2, 0, 0, 0, 0, 1.0, -1.0], np.float32))
  assert obs.shape == (46,) and np.isfinite(obs).all() and np.isfinite(r)
  np.testing.assert_allclose(env.mocap, np.array(MIDPOINT_POS) + [0.01, -0.01, 0.004])     # a * 2.0 * 0.01
  assert (obs[23:] == goal).all()
  # goal_states[0] is "every fixture at rest" = where the env starts: eight solved components (+1 each), success radius 0.3 (kitchen.py:141-183)
  assert s and 7.0 < r <= 8.0
  far = goal.copy(); far[22] = -1.5                                   # microwave wide open as the goal instead
  env.set(INIT_QPOS, np.zeros(23), MIDPOINT_POS, far, INIT_QPOS[:9])
  obs, r, s, _ = env.step(np.zeros(9, np.float32))
  assert not s and r < 0                               # -10 * 1.5 + 7 solved components - 0.5 |mocap - microwave handle|


def test_c_restatement_matches_the_numpy_statement(lm):
  """oracle/physics_oracle.c on the 24-dof table form (joint couplings, dry friction, springs, force limits incl.) vs LinkModel: forward
  accelerations of states with open doors / active couplings / sliding friction, and a 40-timestep env step"""
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  from oracle import physics_c
  cm = physics_c.CModel('kitchen')
  rng = np.random.default_rng(0)
  n = 6
  q = np.tile(INIT_QPOS, (n, 1)) + rng.normal(0, 0.05, (n, 23)); v = rng.normal(0, 0.3, (n, 23)); q[:, 7:9] = rng.uniform(0, 0.04, (n, 2))
  q[1, 22] = -0.5; q[2, 19] = 0.3; q[3, 9] = -0.7; q[3, 10] = -0.004; v[4, 19:] = [3, -2, 2, -4]
  mp = np.tile(MIDPOINT_POS, (n, 1)) + rng.normal(0, 0.05, (n, 3)); mq = np.tile(lm.weld_mocap_quat, (n, 1)); ctrl = rng.uniform(-0.01, 0.05, (n, 2))
  r = cm.run(q, v, mp, mq, ctrl, integrate=False)
  for i in range(n):
    ref = lm.forward(q[i], v[i], ctrl[i], mp[i], mq[i])
    np.testing.assert_allclose(r['qacc'][i], ref['qacc'], rtol=1e-10, atol=1e-10 * np.abs(ref['qacc']).max())
  r2 = cm.run(q[:2], v[:2], mp[:2], mq[:2], ctrl[:2], nsub=40)
  for i in range(2):
    qq, vv, out = q[i].copy(), v[i].copy(), None
    for _ in range(40):                                      # one call = one env step: its first timestep starts the active-set iteration cold
      qq, vv, out = lm.step(qq, vv, ctrl[i], mp[i], mq[i], None if out is None else out['qacc'])
    np.testing.assert_allclose(r2['qpos'][i], qq, atol=1e-12); np.testing.assert_allclose(r2['qvel'][i], vv, atol=1e-11)


def test_task_rows_follow_the_references_rule():
  """tables.npz `kitchen_task_*` (recorded from the reference module by tests/golden/make_golden.py): every named task is the clean goal
  state with its components overwritten (kitchen.py:57-85, convert_to_initial_state), and 'all_pairs' stacks the six pair rows in order"""
  from earl_benchmark_amd import tables
  goal = tables.goal_states('kitchen')[0]
  comp = {'microwave': ([22], [-0.7]), 'light_switch': ([17, 18], [-0.69, -0.05]), 'slide_cabinet': ([19], [0.37]), 'hinge_cabinet': ([20, 21], [0., 1.45])}
  short = {'micro': 'microwave', 'light': 'light_switch', 'slide': 'slide_cabinet', 'hinge': 'hinge_cabinet'}
  assert tables.kitchen_tasks() == sorted(list(comp) + ['micro_hinge', 'micro_slide', 'micro_light', 'light_slide', 'light_hinge', 'slide_hinge', 'all_pairs'])
  for task in tables.kitchen_tasks():
    rows = tables.get(f'kitchen_task_{task}')
    if task == 'all_pairs':
      want = np.stack([tables.get(f'kitchen_task_{t}')[0] for t in ('micro_hinge', 'micro_slide', 'micro_light', 'light_slide', 'light_hinge', 'slide_hinge')])
      np.testing.assert_array_equal(rows, want)
      np.testing.assert_array_equal(rows, tables.initial_states('kitchen'))
      continue
    want = goal.copy()
    for part in ([task] if task in comp else [short[p] for p in task.split('_')]):
      want[comp[part][0]] = comp[part][1]
    assert rows.shape == (1, 23)
    np.testing.assert_array_equal(rows[0], want)


def test_arm_against_static_boxes_collision_tables():
  """Round 3: the declared collision set grew by the hand (eight spheres on link 7), the wrist (two on link 6), the forearm (three on link 5) and the
  finger boxes' corner points against six static boxes taken from the MJCF's own collision geoms (counter-top slab, oven / stove body, back wall,
  hood) or hulls of them (microwave body, cabinet bottoms); the round-2 set is an unchanged PREFIX of the pair / block lists, so states without such
  contacts give the results they gave before."""
  z = np.load(os.path.join(REPO, 'earl_benchmark_amd', 'models', 'kitchen_links.npz'))
  assert len(z['col_pair']) == 393 and len(z['col_blk_begin']) == 61 and len(z['col_box_link']) == 13 and len(z['col_sph_link']) == 92
  assert int(z['col_blk_begin'][26]) == 190 and set(z['col_blk_box'][26:56].tolist()) == set(range(6, 12)) and set(z['col_blk_box'][56:].tolist()) == {12}   # (round 4: the right-hand counter, appended)
  assert sorted(set(z['col_blk_link'][26:].tolist())) == [4, 5, 6, 7, 8]            # forearm, wrist, hand, the two fingers
  new = np.arange(79, 92)
  assert sorted(z['col_sph_link'][new].tolist()) == [4] * 3 + [5] * 2 + [6] * 8
  assert sorted(np.round(z['col_sph_r'][new], 3).tolist()) == [0.02] * 4 + [0.035] * 3 + [0.05] + [0.055] * 2 + [0.06] * 3
  assert (z['col_box_link'][6:] == -1).all()                                     # world-fixed
  m = np.load(os.path.join(REPO, 'earl_benchmark_amd', 'models', 'kitchen.npz'))
  names = [str(x) for x in m['body_names']]
  # the four boxes taken as they stand: size = a colliding box geom of that static body (oven_asset.xml:34-35 etc. via the compiled model)
  for j, body in ((6, 'counters'), (7, 'ovenroot'), (8, 'wallroot'), (9, 'hoodroot')):
    gs = [g for g in range(len(m['geom_body'])) if names[m['geom_body'][g]] == body and m['geom_type'][g] == 4 and (m['geom_contype'][g] or m['geom_conaffinity'][g])]
    assert any(np.allclose(m['geom_size'][g][:3], z['col_box_half'][j]) for g in gs), body
  np.testing.assert_allclose(z['col_box_pos'][6][2] + z['col_box_half'][6][2], 1.60, atol=1e-6)      # counter top
