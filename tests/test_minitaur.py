"""Minitaur on the articulated-body stepper, CPU side (no GPU): the model tables against the facts the reference states, the numpy statement
(oracle/minitaur_oracle.py on oracle/physics_oracle.LinkModel) against first principles, and the C restatement (oracle/physics_oracle.c:
oracle_minitaur_reset / _rollout) against the numpy one.  PARITY WITH THE REFERENCE'S PYBULLET SIMULATION IS UNPINNED AND MODEL-LESS (the URDF is not
in the reference tree); what the reference's own Python computes around Bullet is pinned elsewhere (tests/test_glue.py)."""
import numpy as np
import pytest

from oracle import physics_c
from oracle.minitaur_oracle import GOAL_LOCATIONS, MinitaurOracle
from oracle.physics_oracle import LinkModel, quat_mat


@pytest.fixture(scope='module')
def lm():
  from oracle.minitaur_oracle import MODEL
  return LinkModel(MODEL)


def test_model_facts_the_reference_states(lm):
  """structure, reset pose and loop closures as earl_benchmark/envs/minitaur.py states them (:10-25, :80, :187-217)"""
  assert lm.nv == 22 and len(lm.qpos0) == 23 and int(lm.ball_dof) == 3 and int(lm.weld_att) < 0
  assert list(lm.jtype[:6]) == [1, 1, 1, 2, 3, 3] and (lm.jtype[6:] == 0).all()                 # floating base + 16 revolute joints
  names = [str(x) for x in lm.link_names]
  legs = ('front_left', 'back_left', 'front_right', 'back_right')
  motors = [f'motor_{leg}{s}_joint' for leg in legs for s in 'LR']                                # MOTOR_NAMES, minitaur.py:18-22
  assert [names[int(d)] for d in lm.motor_dof] == motors
  np.testing.assert_array_equal(lm.motor_direction, [-1, -1, -1, -1, 1, 1, 1, 1])                 # :80
  q0 = lm.qpos0
  np.testing.assert_allclose(q0[:7], [0, 0, 0.2, 1, 0, 0, 0])                                     # INIT_POSITION, :10
  for i, (u, w) in enumerate(zip(lm.motor_dof, lm.knee_dof)):
    assert q0[lm.qadr[u]] == lm.motor_direction[i] * np.pi / 2 and q0[lm.qadr[w]] == lm.motor_direction[i] * -2.1834   # :187-211
    assert int(lm.parent[w]) == int(u) and int(lm.parent[u]) == 5
  pos, quat, S = lm.kinematics(q0)
  assert len(lm.con_att1) == 4
  for e in range(4):                                                                              # :12-13, 212-217
    k1, k2 = int(lm.con_att1[e]), int(lm.con_att2[e])
    np.testing.assert_array_equal(lm.att_pos[k1], [0, 0.005, 0.2]); np.testing.assert_array_equal(lm.att_pos[k2], [0, 0.01, 0.2])
    p1, p2 = lm.attachment(pos, quat, k1)[0], lm.attachment(pos, quat, k2)[0]
    assert np.abs(p1 - p2).max() < 1e-12                                                          # the reference's reset angles close every loop
    assert 0.03 < p1[2] < 0.04                                                                    # toes 3.6 cm above the ground at reset
  np.testing.assert_array_equal(lm.gravity, [0, 0, -10]); assert float(lm.timestep) == 0.002      # minitaur_gym_env.py:232, 126-128, 161-164
  # wall tiles (minitaur_assets/wall_tile.urdf:19-24, minitaur_gym_env.py:39-50): inner faces at +-1.5, z in [0.25, 0.75]
  for b in range(1, 5):
    lo, hi = lm.col_box_pos[b] - lm.col_box_half[b], lm.col_box_pos[b] + lm.col_box_half[b]
    assert (lo[2], hi[2]) == (0.25, 0.75) and min(abs(lo[:2]).min(), abs(hi[:2]).min()) == 1.5
  assert 6.0 < lm.mass.sum() < 6.5


def test_free_flight_conserves_momentum_and_the_closures_hold(lm):
  """no gravity, no contacts, no motors, a tumbling start: linear momentum is conserved (the closures are internal forces), the loops stay closed"""
  m = LinkModel({k: getattr(lm, k) for k in vars(lm) if isinstance(getattr(lm, k), np.ndarray)})
  m.gravity = np.zeros(3); m.contacts = False
  rng = np.random.default_rng(3)
  q, v = np.array(m.qpos0, float), np.zeros(22)
  v[:6] = rng.normal(size=6) * [0.3, 0.3, 0.3, 1.0, 1.0, 1.0]
  v[6:] = rng.normal(size=16) * 0.5

  def momentum(q, v):
    pos, quat, S = m.kinematics(q)
    p = np.zeros(3)
    for l in range(22):
      if m.mass[l] > 0:
        c = pos[l] + quat_mat(quat[l]) @ m.com[l]
        V = sum(S[j] * v[j] for j in m.anc[l])
        p += m.mass[l] * (V[3:] + np.cross(V[:3], c))
    return p
  p0 = momentum(q, v)
  for _ in range(200):
    q, v, out = m.step(q, v, np.zeros(0), np.zeros(3), np.array([1.0, 0, 0, 0]))
  p1 = momentum(q, v)
  assert np.abs(p1 - p0).max() < 2e-3 * np.abs(p0).max() + 1e-6
  pos, quat, _ = m.kinematics(q)
  for e in range(4):
    d = m.attachment(pos, quat, int(m.con_att1[e]))[0] - m.attachment(pos, quat, int(m.con_att2[e]))[0]
    assert np.abs(d).max() < 2e-3
  assert abs(np.linalg.norm(q[3:7]) - 1) < 1e-12


def test_reset_stands_and_steps_walk_the_reference_loop():
  o = MinitaurOracle(env_id=2, seed=9)
  obs = o.observation()
  assert obs.shape == (32,) and 0.14 < o.qpos[2] < 0.19 and abs(o.qpos[3]) > 0.999          # standing, upright, after the 100 settle steps
  assert 14.8 <= o.voltage <= 16.8 and 0 <= o.viscous <= 0.01 and any((o.goal == g).all() for g in GOAL_LOCATIONS)
  np.testing.assert_allclose(obs[:8], np.pi / 2, atol=0.25)                                  # the motors hold the commanded pi / 2
  np.testing.assert_array_equal(obs[30:], o.goal); np.testing.assert_array_equal(obs[24:28], o.qpos[[4, 5, 6, 3]])
  with pytest.raises(ValueError):
    o.step(np.full(8, 1.02))                                                                 # minitaur_gym_env.py:276-281
  ob, r, done, info = o.step(np.zeros(8))
  assert done is False and info['success'] in (0.0, 1.0)
  want = 2.0 * (-abs(ob[28] - ob[30]) - abs(ob[29] - ob[31])) - 0.005 * abs(np.dot(ob[8:16], ob[16:24])) * 0.002
  assert abs(r - want) < 1e-15                                                               # _reward == compute_reward(obs) (:505-535)


def test_c_restatement_matches_the_numpy_statement():
  n, T = 3, 8
  c = physics_c.CMinitaur(n, seed=5, env_offset=10)
  oc = c.reset()
  os_ = [MinitaurOracle(env_id=10 + e, seed=5) for e in range(n)]
  on = np.stack([o.observation() for o in os_])
  np.testing.assert_allclose(oc, on, rtol=0, atol=1e-10)
  np.testing.assert_array_equal(c.goal, np.stack([o.goal for o in os_]))
  np.testing.assert_array_equal(c.motor_param, np.stack([[o.voltage, o.viscous, *o.scale, o.foot_mu] for o in os_]))      # the randomizer's six draws: identical bits
  mp = c.motor_param
  assert ((mp[:, 2] > 0.8) & (mp[:, 2] < 1.2)).all() and len(np.unique(mp[:, 2])) == n                                   # SetBaseMass: the model's base x U(0.8, 1.2)
  assert ((mp[:, 3] * 0.275 > 0.8 * 0.275) & (mp[:, 3] * 0.275 < 1.2 * 0.275)).all()                                    # upper link = motor' + leg link'
  assert ((mp[:, 4] * 0.086 > 0.8 * 0.034) & (mp[:, 4] * 0.086 < 1.2 * 0.034)).all()                                    # SetLegMasses' quirk: a lower leg gets the LEG-LINK mass
  assert ((mp[:, 5] > 0.8) & (mp[:, 5] < 1.5)).all()                                                                     # SetFootFriction
  rng = np.random.default_rng(1)
  acts = rng.uniform(-1, 1, (T, n, 8)).astype(np.float32)
  res = c.rollout(acts)
  for t in range(T):
    for e in range(n):
      ob, r, d, info = os_[e].step(acts[t, e])
      np.testing.assert_allclose(res['obs'][t, e], ob, rtol=0, atol=1e-9)
      assert abs(res['reward'][t, e] - r) < 1e-12 and bool(res['success'][t, e]) == bool(info['success'])
  assert (c.steps_since_reset == T).all() and not res['status'].any() and not res['done'].any()
  # overheat protection (minitaur.py:351-358): a motor asked for more than 2.45 N m for more than 500 timesteps is switched off until the next reset
  c2 = physics_c.CMinitaur(1, seed=1, randomize=False, horizon=150)
  c2.reset()
  c2.overheat[:] = 499
  r2 = c2.rollout(np.tile(np.array([1, 1, 1, 1, -1, -1, -1, -1], np.float32), (150, 1, 1)))
  assert r2['done'][-1, 0] and not r2['done'][:-1].any()
  assert np.isfinite(r2['obs']).all()


def test_goal_switch_of_the_lifelong_wrapper_in_the_c_rollout():
  c = physics_c.CMinitaur(2, seed=3, goal_change_frequency=3)
  c.reset()
  g0 = c.goal.copy()
  res = c.rollout(np.zeros((7, 2, 8), np.float32))
  # steps 3 and 6 (indices 2, 5) return the NEW goal in their observation; the reward of that step used the old one (lifelong_wrapper.py:30-44)
  np.testing.assert_array_equal(res['obs'][1, :, 30:], g0)
  o2 = res['obs'][2]
  r_old = 2.0 * (-np.abs(o2[:, 28] - g0[:, 0]) - np.abs(o2[:, 29] - g0[:, 1])) - 0.005 * np.abs((o2[:, 8:16] * o2[:, 16:24]).sum(1)) * 0.002
  np.testing.assert_allclose(res['reward'][2], r_old, rtol=0, atol=1e-12)
  np.testing.assert_array_equal(res['obs'][6, :, 30:], c.goal)
  assert (c.steps_since_goal_change == 1).all()


def test_two_shards_equal_one_batch_on_the_cpu_statement():
  """env-range sharding: Philox streams are keyed by the GLOBAL env id, so two half shards (env_offset) reproduce the batch bit for bit --
  reset draws (goal, battery voltage, viscous damping) and rollouts alike"""
  from earl_benchmark_amd import sharding
  n, T = 6, 4
  full = physics_c.CMinitaur(n, seed=21)
  full.reset()
  rng = np.random.default_rng(2)
  acts = rng.uniform(-1, 1, (T, n, 8)).astype(np.float32)
  rf = full.rollout(acts)
  parts = []
  for rank in range(2):
    kw = sharding.shard_kwargs(n, rank, 2)
    sh = physics_c.CMinitaur(kw['num_envs'], seed=21, env_offset=kw['env_offset'])
    sh.reset()
    lo = kw['env_offset']
    np.testing.assert_array_equal(sh.goal, full.goal[lo:lo + kw['num_envs']])
    np.testing.assert_array_equal(sh.motor_param, full.motor_param[lo:lo + kw['num_envs']])
    parts.append(sh.rollout(acts[:, lo:lo + kw['num_envs']]))
  np.testing.assert_array_equal(np.concatenate([p['obs'] for p in parts], 1), rf['obs'])
  np.testing.assert_array_equal(np.concatenate([p['reward'] for p in parts], 1), rf['reward'])


def test_carried_edge_sets_change_the_number_of_passes_not_the_results():
  """Start of the active-set passes (oracle/physics_oracle.c StepOut.pact, csrc/minitaur_stepper.h C3): a contact slot that holds the same collision pair as at the timestep
  before starts from the edge set its passes ended with.  The fixed point is the same -- outputs and state bit for bit -- and it is reached in fewer passes."""
  import ctypes as C
  lib = physics_c.lib()
  n, T = 48, 40
  acts = np.random.default_rng(5).uniform(-1, 1, (T, n, 8)).astype(np.float32)
  runs = {}
  try:
    for carry in (1, 0):
      lib.oracle_set_carry_sets(C.c_int(carry))
      c = physics_c.CMinitaur(n, seed=1234)
      c.reset()
      st = (C.c_longlong * 5)()
      lib.oracle_newton_stats(st, C.c_int(1))
      res = c.rollout(acts)
      lib.oracle_newton_stats(st, C.c_int(1))
      runs[carry] = (res, c.qpos.copy(), c.qvel.copy(), st[1] / st[0], st[4])
  finally:
    lib.oracle_set_carry_sets(C.c_int(1))
  for k in ('obs', 'reward', 'done', 'success', 'status'):
    np.testing.assert_array_equal(runs[1][0][k], runs[0][0][k])
  np.testing.assert_array_equal(runs[1][1], runs[0][1])
  np.testing.assert_array_equal(runs[1][2], runs[0][2])
  assert runs[1][4] == 0 and runs[0][4] == 0                      # every timestep reached its fixed point
  assert runs[1][3] < 0.9 * runs[0][3], (runs[1][3], runs[0][3])   # 1.66 against 2.05 passes per timestep on 3 M timesteps of random actions
