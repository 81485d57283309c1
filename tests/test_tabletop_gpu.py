"""Parity of the HIP tabletop path (through the C ABI) with the CPU oracle and the golden vectors.

Bar: bit-exact for fp64 state, f32 observations, attached flag, done, success, sparse reward and every integer
counter; dense reward within DENSE tolerance (device exp() vs libm: <= 1 ulp of fp64 before the f32 rounding).
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

DENSE_RTOL, DENSE_ATOL = 1e-6, 1e-6          # HIP vs oracle (same formula, different libm)
GOLD_RTOL, GOLD_ATOL = 2e-6, 2e-6            # vs goldens recorded under numpy 2 (f32 evaluation of the formula)


@pytest.fixture(scope='module')
def hx():
  import hip_harness
  return hip_harness


@pytest.fixture(scope='module')
def orc():
  from oracle import tabletop_oracle
  return tabletop_oracle


def assert_same_state(o, h):
  for k in h.STATE:
    np.testing.assert_array_equal(h.host(k), getattr(o, k), err_msg=k)


def assert_same_out(a, b, dense):
  obs_a, rew_a, done_a, succ_a = a
  obs_b, rew_b, done_b, succ_b = b
  np.testing.assert_array_equal(obs_a.view(np.uint32), obs_b.view(np.uint32))   # bit pattern, NaN/-0.0 included
  np.testing.assert_array_equal(done_a, done_b)
  np.testing.assert_array_equal(succ_a, succ_b)
  if dense:
    np.testing.assert_allclose(rew_a, rew_b, rtol=DENSE_RTOL, atol=DENSE_ATOL)
  else:
    np.testing.assert_array_equal(rew_a, rew_b)


def np122(returned, norm, radius=0.2):
  return np.where(norm == np.float32(radius), False, returned.astype(bool))


# ------------------------------------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize('rt', ['sparse', 'dense'])
def test_onestep_golden(hx, orc, rt):
  g = load_golden('tabletop_onestep')
  n = len(g['qpos0'])
  o = orc.OracleTabletop(n, reward_type=rt, horizon=10**9, goal_table=g['goal'])
  o.goal_idx[:] = np.arange(n); o.qpos[:] = g['qpos0']; o.attached[:] = g['attached0']
  h = hx.HipTabletop(n, reward_type=rt, horizon=10**9, goal_table=g['goal'])
  h.set_from(o)
  out_h = h.step(g['action'])
  out_o = o.step(g['action'])
  assert_same_out(out_h, out_o, rt == 'dense')
  assert_same_state(o, h)
  obs, rew, done, succ = out_h
  np.testing.assert_array_equal(h.host('qpos'), g['qpos1'])
  np.testing.assert_array_equal(h.host('attached'), g['attached1'])
  np.testing.assert_array_equal(obs, g['obs'])
  np.testing.assert_array_equal(succ.astype(bool), g['success'])
  if rt == 'sparse':
    np.testing.assert_array_equal(rew, g['reward_sparse'])
  else:
    np.testing.assert_allclose(rew, g['reward_dense'], rtol=GOLD_RTOL, atol=GOLD_ATOL)


def test_wide_success_and_pure_reward_golden(hx):
  g = load_golden('tabletop_onestep')
  r, s = hx.hip_reward(g['obs'], 'sparse')
  np.testing.assert_array_equal(r, g['reward_sparse']); np.testing.assert_array_equal(s.astype(bool), g['success'])
  r, s = hx.hip_reward(g['obs'], 'sparse', wide_init=True)
  np.testing.assert_array_equal(r, g['reward_sparse_wide']); np.testing.assert_array_equal(s.astype(bool), g['success_wide'])
  r, s = hx.hip_reward(g['obs'], 'dense')
  np.testing.assert_allclose(r, g['reward_dense'], rtol=GOLD_RTOL, atol=GOLD_ATOL)


@pytest.mark.parametrize('rt', ['sparse', 'dense'])
def test_rollouts_golden(hx, rt):
  g = load_golden('tabletop_rollouts')
  acts = np.ascontiguousarray(g['actions'].transpose(1, 0, 2))
  T, R, _ = acts.shape
  h = hx.HipTabletop(R, reward_type=rt, horizon=int(g['horizon']))
  obs0 = h.reset(next_goal_idx=g['goal_idx'])
  np.testing.assert_array_equal(obs0, g[f'{rt}_obs0'])
  obs, rew, done, succ = h.rollout(acts)
  want_succ = np122(g[f'{rt}_success'], g[f'{rt}_norm4'])
  np.testing.assert_array_equal(obs.transpose(1, 0, 2), g[f'{rt}_obs'])
  np.testing.assert_array_equal(done.T.astype(bool), g[f'{rt}_done'])
  np.testing.assert_array_equal(succ.T.astype(bool), want_succ)
  np.testing.assert_array_equal(h.host('qpos'), g[f'{rt}_qpos'][:, -1])
  if rt == 'sparse':
    np.testing.assert_array_equal(rew.T, want_succ.astype(np.float32))
  else:
    np.testing.assert_allclose(rew.T, g[f'{rt}_reward'], rtol=GOLD_RTOL, atol=GOLD_ATOL)
  # the same through T single-step launches
  h2 = hx.HipTabletop(R, reward_type=rt, horizon=int(g['horizon']))
  h2.reset(next_goal_idx=g['goal_idx'])
  for t in range(T):
    ob, rw, dn, sc = h2.step(acts[t])
    np.testing.assert_array_equal(ob, obs[t]); np.testing.assert_array_equal(rw, rew[t])
    np.testing.assert_array_equal(dn, done[t]); np.testing.assert_array_equal(sc, succ[t])
  for k in h.STATE:
    np.testing.assert_array_equal(h.host(k), h2.host(k))


@pytest.mark.parametrize('direction', ['forward', 'reverse'])
def test_demonstrations_replay(hx, direction):
  import os
  from conftest import REPO
  demo = np.load(os.path.join(REPO, 'earl_benchmark_amd', 'demonstrations', 'tabletop_manipulation', direction, 'demo_data.npz'))
  rep = load_golden('tabletop_demo_replay')
  ob, act = demo['observations'], demo['actions']
  n = len(ob)
  h = hx.HipTabletop(n, horizon=10**9, goal_table=ob[:, 6:12].astype(np.float64))
  h.goal_idx.copy_(__import__('torch').arange(n))
  h.qpos.copy_(__import__('torch').from_numpy(ob[:, :4].astype(np.float64)))
  h.attached.copy_(__import__('torch').from_numpy(ob[:, 4].astype(np.int8)))
  obs, rew, done, succ = h.step(act)
  assert np.abs(obs - demo['next_observations']).max() < 5e-7
  np.testing.assert_array_equal(rew, demo['rewards'][:, 0])
  np.testing.assert_array_equal(obs, rep[f'{direction}_next_obs'])
  np.testing.assert_array_equal(h.host('attached'), rep[f'{direction}_attached'])


def test_wide_init_golden(hx):
  g = load_golden('tabletop_wide_init')
  np.testing.assert_array_equal(hx.hip_valid_init(g['candidates']).astype(bool), g['valid'])


@pytest.mark.parametrize('rt', ['sparse', 'dense'])
def test_lifelong_golden(hx, rt):
  g = load_golden('tabletop_lifelong')
  T = len(g[f'{rt}_actions'])
  h = hx.HipTabletop(1, reward_type=rt, horizon=int(g['train_horizon']), goal_change_frequency=int(g['freq']))
  obs0 = h.reset(next_goal_idx=[int(g[f'{rt}_goal0'])])
  np.testing.assert_array_equal(obs0[0], g[f'{rt}_obs0'])
  for t in range(T):
    ob, rw, dn, sc = h.step(g[f'{rt}_actions'][t][None], next_goal_idx=[int(g[f'{rt}_goal_seq'][t])])
    np.testing.assert_array_equal(ob[0], g[f'{rt}_obs'][t])
    assert bool(dn[0]) == bool(g[f'{rt}_done'][t])
    ret = h.host('lifelong_return')[0]
    if rt == 'sparse':
      assert rw[0] == g[f'{rt}_reward'][t] and ret == g[f'{rt}_return'][t]
    else:
      np.testing.assert_allclose(rw[0], g[f'{rt}_reward'][t], rtol=GOLD_RTOL, atol=GOLD_ATOL)
      np.testing.assert_allclose(ret, g[f'{rt}_return'][t], rtol=1e-5)


# ------------------------------------------------------------------------------------------------ vs the oracle
def random_state(rng, o):
  n = o.n
  q = rng.uniform(-2.8, 2.8, size=(n, o.nq))
  near = rng.random(n) < 0.4
  for k in range(o.nobj):
    r = rng.uniform(0, 0.8, size=n); th = rng.uniform(0, 2 * np.pi, size=n)
    q[near, 2 + 2 * k] = (q[:, 0] + r * np.cos(th))[near]
    q[near, 3 + 2 * k] = (q[:, 1] + r * np.sin(th))[near]
  o.qpos[:] = np.clip(q, -2.8, 2.8)
  o.attached[:] = np.where(rng.random(n) < 0.3, rng.integers(0, o.nobj, size=n), -1)
  o.goal_idx[:] = rng.integers(0, len(o.goal_table), size=n)


@pytest.mark.parametrize('rt,wide', [('sparse', False), ('dense', False), ('sparse', True)])
@pytest.mark.parametrize('n', [1, 63, 257, 4096, 100003])
def test_step_matches_oracle_random(hx, orc, rt, wide, n):
  rng = np.random.default_rng(n)
  o = orc.OracleTabletop(n, reward_type=rt, wide_init=wide, horizon=3)
  random_state(rng, o)
  h = hx.HipTabletop(n, reward_type=rt, wide_init=wide, horizon=3)
  h.set_from(o)
  for t in range(4):
    act = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    assert_same_out(h.step(act), o.step(act), rt == 'dense')
    assert_same_state(o, h)


def test_empty_batch_and_bad_args(hx):
  import ctypes as C
  from earl_benchmark_amd import _abi
  h = hx.HipTabletop(4)
  h.cfg.n = 0
  st = h._state()
  _, out = h._outs((4,))
  assert h.lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), h.qpos.data_ptr(), None, C.byref(out), None) == 0
  h.cfg.n = 4
  assert h.lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), None, None, C.byref(out), None) == -1
  assert b'NULL' in h.lib.earl_last_error()
  h.cfg.reward_type = 5
  assert h.lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), h.qpos.data_ptr(), None, C.byref(out), None) == -1
  h.cfg.reward_type = 0
  h.cfg.goal_change_frequency = 3
  st.lifelong_return = None
  assert h.lib.earl_tabletop_step(C.byref(h.cfg), C.byref(st), h.qpos.data_ptr(), None, C.byref(out), None) == -1
  with pytest.raises(_abi.EarlHipError):
    _abi.check(-1, 'x')


def test_special_values(hx, orc):
  """NaN / inf / -0.0 / out-of-range actions, grip decided on the RESCALED action (App. B3), walls (B6)."""
  specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-17, 1.2e-16, 2.0 ** -53, -1e-30, 1.0, -1.0, 5.0, -5.0, 0.5, 0.1, -0.7],
                      np.float32)
  acts = np.array([[a, b, c] for a in specials for b in specials[:6] for c in specials], np.float32)
  n = len(acts)
  o = orc.OracleTabletop(n, horizon=10**9)
  rng = np.random.default_rng(0)
  random_state(rng, o)
  o.qpos[::5] = [2.7, 0.0, 2.75, 0.0]          # object catches up to the wall
  o.attached[::5] = 0
  h = hx.HipTabletop(n, horizon=10**9)
  h.set_from(o)
  assert_same_out(h.step(acts), o.step(acts), False)
  assert_same_state(o, h)
  # one more step from the (partly NaN) state
  assert_same_out(h.step(acts[::-1].copy()), o.step(acts[::-1].copy()), False)
  assert_same_state(o, h)


def test_threshold_boundaries(hx, orc):
  """The kernels compare squared distances against precomputed exact thresholds instead of taking square roots;
  walk the neighbourhood of every threshold ulp by ulp and compare with the oracle (which takes the roots)."""
  # grasp radius: gripper at origin, object at distance ~0.4 along an axis and along a diagonal
  cands = []
  for base in (0.4, np.sqrt(0.08)):
    x = base
    for _ in range(40):
      x = np.nextafter(x, 0)
    for _ in range(80):
      cands.append(x); x = np.nextafter(x, 1)
  n = 2 * len(cands)
  o = orc.OracleTabletop(n, horizon=10**9)
  o.qpos[:] = 0
  o.qpos[:len(cands), 2] = cands
  o.qpos[len(cands):, 2] = cands[:len(cands)]; o.qpos[len(cands):, 3] = cands[:len(cands)]
  h = hx.HipTabletop(n, horizon=10**9)
  h.set_from(o)
  act = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
  assert_same_out(h.step(act), o.step(act), False)
  assert_same_state(o, h)
  assert 0 < (o.attached == 0).sum() < n
  # success radius on f32 observations: craft obs whose f32 norm walks through float32(0.2)
  obs = np.zeros((4000, 12), np.float32)
  x = np.float32(0.2)
  for _ in range(1000):
    x = np.nextafter(x, np.float32(0))
  for i in range(2000):
    obs[i, 0] = x; obs[2000 + i, 2] = x; x = np.nextafter(x, np.float32(1))
  obs[:, 4:6] = -1; obs[:, 10:] = -1
  for wide in (False, True):
    r, s = hx.hip_reward(obs, 'sparse', wide_init=wide)
    r0, _, s0 = orc.reward(obs, 'sparse', wide_init=wide)
    np.testing.assert_array_equal(s, s0); np.testing.assert_array_equal(r, r0)
  assert 0 < s0.sum() < 4000
  # valid-init radius 1
  c = []
  x = 1.0
  for _ in range(30):
    x = np.nextafter(x, 0)
  for _ in range(60):
    c.append([x, 0, 0, 0]); c.append([2.0, 2.0, -2.5 + x, -1.0]); x = np.nextafter(x, 2)
  c = np.array(c)
  np.testing.assert_array_equal(hx.hip_valid_init(c), orc.valid_init(c))


@pytest.mark.parametrize('mode', ['fixed', 'at_goal', 'wide'])
def test_reset_matches_oracle(hx, orc, mode):
  n = 5000
  kw = dict(reset_at_goal=mode == 'at_goal', wide_init=mode == 'wide', seed=1234, env_offset=77)
  o = orc.OracleTabletop(n, **kw)
  h = hx.HipTabletop(n, **kw)
  np.testing.assert_array_equal(h.reset(), o.reset())
  assert_same_state(o, h)
  rng = np.random.default_rng(1)
  random_state(rng, o)
  o.steps_since_reset[:] = 9
  h.set_from(o)
  mask = (rng.random(n) < 0.5).astype(np.uint8)
  np.testing.assert_array_equal(h.reset(mask=mask), o.reset(mask=mask))
  assert_same_state(o, h)
  inj = rng.integers(0, 4, size=n).astype(np.int32)
  np.testing.assert_array_equal(h.reset(next_goal_idx=inj), o.reset(next_goal_idx=inj))
  assert_same_state(o, h)
  assert (o.goal_idx == inj).all()


@pytest.mark.parametrize('rt', ['sparse', 'dense'])
def test_rollout_autoreset_and_lifelong_match_oracle(hx, orc, rt):
  n, T = 777, 64
  rng = np.random.default_rng(5)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  acts[..., 2] = np.abs(acts[..., 2])
  for kw in (dict(horizon=10, auto_reset=True, wide_init=True), dict(horizon=10, auto_reset=True, reset_at_goal=True),
             dict(horizon=1000, goal_change_frequency=7), dict(horizon=17)):
    o = orc.OracleTabletop(n, reward_type=rt, seed=99, **kw)
    h = hx.HipTabletop(n, reward_type=rt, seed=99, **kw)
    np.testing.assert_array_equal(h.reset(), o.reset())
    assert_same_out(h.rollout(acts), o.rollout(acts), rt == 'dense')
    if rt == 'dense' and kw.get('goal_change_frequency'):
      np.testing.assert_allclose(h.host('lifelong_return'), o.lifelong_return, rtol=1e-9)
      o.lifelong_return[:] = h.host('lifelong_return')
    assert_same_state(o, h)
    # and step by step on the device: identical to the fused launch
    h2 = hx.HipTabletop(n, reward_type=rt, seed=99, **kw)
    h2.reset()
    for t in range(T):
      h2.step(acts[t])
    for k in h.STATE:
      np.testing.assert_array_equal(h.host(k), h2.host(k), err_msg=k)


def test_sharding_invariance(hx):
  """RNG streams are keyed by the global env id: two half shards == one full batch."""
  n, T = 1024, 40
  rng = np.random.default_rng(8)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  kw = dict(horizon=9, auto_reset=True, wide_init=True, seed=4)
  full = hx.HipTabletop(n, **kw)
  a = hx.HipTabletop(n // 2, **kw)
  b = hx.HipTabletop(n // 2, env_offset=n // 2, **kw)
  obs0 = full.reset()
  np.testing.assert_array_equal(np.concatenate([a.reset(), b.reset()]), obs0)
  out = full.rollout(acts)
  oa = a.rollout(np.ascontiguousarray(acts[:, :n // 2])); ob = b.rollout(np.ascontiguousarray(acts[:, n // 2:]))
  for x, xa, xb in zip(out, oa, ob):
    np.testing.assert_array_equal(x, np.concatenate([xa, xb], axis=1))


# ------------------------------------------------------------------------------------------------ 3-object variant
def test_3obj_golden_and_oracle(hx, orc):
  g = load_golden('tabletop3_onestep')
  n = len(g['qpos0'])
  for rt in ('sparse', 'dense'):
    o = orc.OracleTabletop(n, reward_type=rt, horizon=10**9, goal_table=g['goal'], nobj=3)
    o.goal_idx[:] = np.arange(n); o.qpos[:] = g['qpos0']; o.attached[:] = g['attached0']
    h = hx.HipTabletop(n, reward_type=rt, horizon=10**9, goal_table=g['goal'], nobj=3)
    h.set_from(o)
    out = h.step(g['action'])
    assert_same_out(out, o.step(g['action']), rt == 'dense')
    assert_same_state(o, h)
    np.testing.assert_array_equal(out[0], g['obs'])
    np.testing.assert_array_equal(h.host('qpos'), g['qpos1'])
    np.testing.assert_array_equal(h.host('attached'), g['attached1'])
    np.testing.assert_array_equal(out[3].astype(bool), g['success'])
    acts = np.ascontiguousarray(g['roll_actions'].transpose(1, 0, 2))
    h = hx.HipTabletop(acts.shape[1], reward_type=rt, horizon=10**9, nobj=3)
    np.testing.assert_array_equal(h.reset(), g[f'roll_{rt}_obs0'])
    obs, rew, done, succ = h.rollout(acts)
    np.testing.assert_array_equal(obs.transpose(1, 0, 2), g[f'roll_{rt}_obs'])
    if rt == 'sparse':
      np.testing.assert_array_equal(rew.T, g[f'roll_{rt}_reward'].astype(np.float32))
    else:
      np.testing.assert_allclose(rew.T, g[f'roll_{rt}_reward'], rtol=GOLD_RTOL, atol=GOLD_ATOL)
    r, s = hx.hip_reward(g['obs'], rt, nobj=3)
    r0, _, s0 = orc.reward(g['obs'], rt, nobj=3)
    np.testing.assert_array_equal(s, s0)
    if rt == 'sparse':
      np.testing.assert_array_equal(r, r0)
    else:
      np.testing.assert_allclose(r, r0, rtol=DENSE_RTOL, atol=DENSE_ATOL)


def test_3obj_closest_object_ties(hx, orc):
  """Two objects whose squared distances differ by an ulp can have EQUAL rounded distances: the first must win."""
  n = 4096
  rng = np.random.default_rng(3)
  o = orc.OracleTabletop(n, horizon=10**9, nobj=3)
  o.qpos[:] = 0
  d = rng.uniform(0.05, 0.39, size=n)
  o.qpos[:, 4] = d                                   # object 1 at distance d
  o.qpos[:, 2] = np.nextafter(d, 1)                  # object 0 one ulp farther
  o.qpos[:, 7] = np.nextafter(np.nextafter(d, 0), 0) # object 2 two ulps nearer
  o.qpos[::2, 7] = 2.0
  h = hx.HipTabletop(n, horizon=10**9, nobj=3)
  h.set_from(o)
  act = np.tile(np.array([0.3, -0.2, 1], np.float32), (n, 1))
  assert_same_out(h.step(act), o.step(act), False)
  assert_same_state(o, h)
  assert len(np.unique(o.attached)) >= 2


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties(hx):
  """BASELINE config: 4096 envs x 200 steps.  Size-independent properties instead of a stored answer:
  determinism, sharding-invariance of a checksum, walls, latch semantics, done exactly at the horizon."""
  n, T = 4096, 200
  rng = np.random.default_rng(11)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  h = hx.HipTabletop(n, horizon=T, seed=2)
  h.reset()
  obs, rew, done, succ = h.rollout(acts)
  assert np.isfinite(obs).all() and (np.abs(obs[..., :4]) <= 2.8).all()
  assert not done[:-1].any() and done[-1].all()
  np.testing.assert_array_equal(rew, succ.astype(np.float32))
  # the gripper is the clipped cumulative sum of rescaled actions
  a = -0.2 + (np.clip(acts[..., :2].astype(np.float64), -1, 1) + 1.) * 0.5 * 0.4
  f = np.zeros((n, 2))
  for t in range(T):
    f = np.clip(f + a[t], -2.8, 2.8)
    np.testing.assert_array_equal(obs[t, :, :2], f.astype(np.float32))
  # attached flag can only turn on while grip > 0, and is off whenever grip <= 0
  grip = a_grip = -0.2 + (np.clip(acts[..., 2].astype(np.float64), -1, 1) + 1.) * 0.5 * 0.4
  assert (obs[..., 4][grip <= 0] == -1).all()
  # determinism
  h2 = hx.HipTabletop(n, horizon=T, seed=2)
  h2.reset()
  out2 = h2.rollout(acts)
  for x, y in zip((obs, rew, done, succ), out2):
    np.testing.assert_array_equal(x, y)


# ------------------------------------------------------------------------------------------------ kernel variants
@pytest.mark.parametrize('rt,wide', [('sparse', False), ('dense', False), ('sparse', True)])
@pytest.mark.parametrize('n,T', [(1, 1), (63, 5), (64, 4), (65, 13), (1000, 37), (4096, 200), (10001, 9)])
def test_rollout_kernels_agree(hx, orc, rt, wide, n, T):
  """The wave-specialised rollout kernel (default) == the plain one-lane-per-env kernel == the oracle, including
  partial workgroups, n % 4 != 0 (byte-wise flag stores), T not a multiple of the chunk, NaN/inf actions."""
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  rng = np.random.default_rng(n * 1000 + T)
  acts = rng.uniform(-1.3, 1.3, size=(T, n, 3)).astype(np.float32)
  acts[..., 2] = np.where(rng.random((T, n)) < 0.7, np.abs(acts[..., 2]), acts[..., 2])
  if n >= 63:
    acts[T // 2, 5, 0] = np.nan          # poisons env 5 from step T//2 on (np.clip propagates NaN)
    acts[0, 7, 1] = np.inf
    acts[T - 1, 11, 2] = np.nan          # NaN grip -> release
  o = orc.OracleTabletop(n, reward_type=rt, wide_init=wide, horizon=max(1, T - 2), seed=1)
  random_state(rng, o)
  o.steps_since_reset[:] = rng.integers(0, 3, size=n)
  outs = {}
  # 0 = shipped configuration, 1 = plain kernel; sparse only: 11 = x / y in the two lane halves (v_permlane32_swap), 13 = one lane per
  # env with VGPR-only masks, 20 / 22 = x / y in adjacent lanes (DPP) with VGPR-only masks and other role counts
  impls = (0, 1) + ((11, 13, 20, 22) if rt == 'sparse' else ())
  for impl in impls:
    h = hx.HipTabletop(n, reward_type=rt, wide_init=wide, horizon=max(1, T - 2), seed=1)
    h.set_from(o)
    prev = lib.earl_debug_set_rollout_impl(impl)
    try:
      outs[impl] = h.rollout(acts)
    finally:
      lib.earl_debug_set_rollout_impl(prev)
    outs[impl, 'state'] = {k: h.host(k) for k in h.STATE}
  for impl in impls[1:]:
    for x, y in zip(outs[0], outs[impl]):
      np.testing.assert_array_equal(x.view(np.uint8), y.view(np.uint8), err_msg=f'variant {impl}')
    for k in outs[0, 'state']:
      np.testing.assert_array_equal(outs[0, 'state'][k], outs[impl, 'state'][k], err_msg=f'{k} variant {impl}')
  want = o.rollout(acts)
  assert_same_out(outs[0], want, rt == 'dense')
  for k, v in outs[0, 'state'].items():
    np.testing.assert_array_equal(v, getattr(o, k), err_msg=k)
  if n >= 63 and T >= 4:
    assert np.isnan(outs[0][0][-1, 5, 0])


@pytest.mark.parametrize('mode', ['fixed', 'at_goal', 'wide'])
@pytest.mark.parametrize('impl', [0, 1])
def test_fused_reset_rollout_equals_reset_then_rollout(hx, orc, mode, impl):
  """earl_tabletop_reset_rollout (one launch) == reset + rollout (two launches) == the oracle, for every reset mode,
  for both rollout kernels, twice in a row (second episode starts from a dirty state)."""
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  n, T = 1000, 21
  kw = dict(reset_at_goal=mode == 'at_goal', wide_init=mode == 'wide', seed=77, env_offset=5, horizon=T)
  rng = np.random.default_rng(2)
  o = orc.OracleTabletop(n, **kw)
  h = hx.HipTabletop(n, **kw)
  prev = lib.earl_debug_set_rollout_impl(impl)
  try:
    for rep in range(2):
      acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
      acts[..., 2] = np.abs(acts[..., 2])
      got = h.rollout(acts, reset_first=True)
      o.reset()
      want = o.rollout(acts)
      assert_same_out(got, want, False)
      assert_same_state(o, h)
      assert h.cfg.counter == o.cfg.counter
  finally:
    lib.earl_debug_set_rollout_impl(prev)
  assert (o.num_interventions == 2).all() and (o.steps_since_reset == T).all()


def test_3obj_reset_at_goal(hx, orc):
  """3obj reset_at_goal (:64-69): goal + U(-0.3, 0.3)^8 -- HIP == oracle bit for bit, noise inside the box."""
  n = 3000
  kw = dict(reset_at_goal=True, nobj=3, seed=21, env_offset=9, horizon=50)
  o = orc.OracleTabletop(n, **kw)
  h = hx.HipTabletop(n, **kw)
  np.testing.assert_array_equal(h.reset(), o.reset())
  assert_same_state(o, h)
  d = o.qpos - orc.GOAL_TABLE3[0, :8]
  assert (np.abs(d) <= 0.3).all() and d.std() > 0.15 and abs(d.mean()) < 0.01
  m = (np.arange(n) % 3 == 0).astype(np.uint8)
  np.testing.assert_array_equal(h.reset(mask=m), o.reset(mask=m))
  assert_same_state(o, h)


@pytest.mark.parametrize('n,T,E,kw', [(4096, 200, 3, {}), (257, 24, 4, {'wide_init_distr': True}), (300, 16, 2, {'reset_at_goal': True}),
                                      (130, 40, 3, {'reward_type': 'dense'}), (100, 20, 3, {}), (64, 8, 2, {}),
                                      # 16-step chunks (T >= 32, up to 256 workgroups): episodes that end in the middle of a chunk (T % 16 == 8), a launch
                                      # that ends in half a chunk (odd E x such a T), episodes of exactly two chunks, the shortest ones the path takes
                                      (4096, 200, 4, {}), (1000, 40, 5, {}), (200, 32, 3, {'wide_init_distr': True}), (333, 56, 7, {'reward_type': 'dense'}),
                                      (129, 48, 2, {'reset_at_goal': True}), (16384, 72, 3, {})])
def test_eval_episodes_in_one_launch_equal_the_sequence_of_launches(n, T, E, kw):
  """earl_tabletop_eval_episodes: E x (reset + T steps) in ONE launch of the wave-specialised kernel (episode boundaries inside the
  launch: state reset, goal re-drawn, goal part of the row images rewritten, done counted per episode) == E fused reset+rollout launches,
  bit for bit, outputs and final state; also through the fallback (T not a multiple of the 8-step chunk, T < 16)."""
  import torch
  from earl_benchmark_amd.envs import tabletop
  kw = dict({'reward_type': 'sparse'}, **kw)
  a = tabletop.TabletopManipulation(num_envs=n, seed=13, scalar_api=False, **kw)
  b = tabletop.TabletopManipulation(num_envs=n, seed=13, scalar_api=False, **kw)
  a._cfg.horizon = b._cfg.horizon = T
  g = torch.Generator(device='cuda').manual_seed(4)
  acts = (torch.rand(E, T, n, 3, generator=g, device='cuda') * 2 - 1).contiguous()
  acts[..., 2] = acts[..., 2].abs() * (torch.rand(E, T, n, generator=g, device='cuda') > 0.2)       # mostly gripping: attach / carry / release
  for shared in (False, True):
    got = a.rollout_episodes(acts[0], episodes=E) if shared else a.rollout_episodes(acts)
    for e in range(E):
      want = b.rollout(acts[0] if shared else acts[e], reset_first=True)
      for x, y, name in zip(got, want, ('obs', 'reward', 'done', 'success')):
        assert torch.equal(x[e], y), (shared, e, name)
    for k in ('qpos', 'attached', 'goal_idx', 'steps_since_reset', 'interventions'):
      assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert a._cfg.counter == b._cfg.counter and a.total_step_count == b.total_step_count
    assert bool(got[2][:, -1].all()) and not bool(got[2][:, :-1].any())            # done exactly at the last step of EVERY episode
  assert len(torch.unique(a.goal_idx)) > 1                                          # goals were re-drawn per env


@pytest.mark.parametrize('n,T,E,kw', [
    # the launches bench.py times (4096 envs = 64 workgroups -> four episode groups): default flags (28 = 4 x 7), the driver's
    # --steps 20 (4 x 5), a ragged split (9 -> 3 groups x 3) and a small batch with many groups whose last one is short (37 -> 13 x 3, last 1)
    (4096, 200, 28, {}), (4096, 200, 20, {}), (4096, 200, 9, {}), (1000, 40, 37, {}),
    # same launch geometry through the other instantiations: dense reward, wide init, reset at goal, 8-step chunks (T < 32), two workgroups per group
    (4096, 200, 10, {'reward_type': 'dense'}), (700, 48, 11, {'wide_init': True}), (333, 24, 10, {'reset_at_goal': True}), (8192, 56, 6, {})])
def test_eval_episodes_groups_of_several_episodes_against_the_oracle(hx, orc, n, T, E, kw):
  """The launch shape the bench times: earl_tabletop_eval_episodes with SEVERAL episode groups side by side AND several episodes per group
  (csrc/tabletop.hip do_rollout: ep_groups > 1, ep_per_group > 1), every episode with its OWN actions -- compared episode by episode, bit for
  bit, with the ORACLE walking the reference's evaluation loop (persistent_state_wrapper.py:17-31: reset(), then T steps) with the counters of
  the sequence; final state and wrapper counters too."""
  import torch
  kw = dict({'reward_type': 'sparse'}, **kw)
  base = dict(seed=31, env_offset=11, horizon=T, **kw)
  o = orc.OracleTabletop(n, **base)
  h = hx.HipTabletop(n, **base)
  rng = np.random.default_rng(n + T + E)
  o.qpos[:] = rng.uniform(-2.8, 2.8, size=o.qpos.shape)                      # a dirty state left by earlier use: every episode must start from its reset
  o.attached[:] = rng.integers(-1, 1, size=n)
  o.steps_since_reset[:] = rng.integers(0, T, size=n)
  o.num_interventions[:] = rng.integers(0, 5, size=n)
  o.cfg.counter = 1000
  h.set_from(o)
  g = torch.Generator(device='cuda').manual_seed(E * 1000 + T)
  acts = (torch.rand(E, T, n, 3, generator=g, device='cuda') * 2 - 1).contiguous()
  acts[..., 2] = acts[..., 2].abs() * (torch.rand(E, T, n, generator=g, device='cuda') > 0.15)     # mostly gripping: attach / carry / release
  acts[..., :2] *= 1.0 + 2.0 * (torch.rand(E, T, n, 1, generator=g, device='cuda') > 0.9)          # some moves beyond the action box (clipped)
  got = h.eval_episodes(acts)
  dense = kw['reward_type'] == 'dense'
  goals = set()
  for e in range(E):                                                         # every episode, not a sample: the oracle walks 23 M env-steps in seconds
    a = acts[e].cpu().numpy()
    o.reset()
    goals.update(np.unique(o.goal_idx).tolist())
    want = o.rollout(a)
    assert_same_out(tuple(x[e].cpu().numpy() for x in got), want, dense)
    assert want[2][-1].all() and not want[2][:-1].any()                      # done exactly at the horizon, every episode
  assert_same_state(o, h)
  assert h.cfg.counter == o.cfg.counter
  assert len(goals) == 4 and (E < 3 or not torch.equal(got[0][0], got[0][E - 1]))


@pytest.mark.parametrize('case', range(24))
def test_eval_episodes_fuzz_against_the_oracle(hx, orc, case):
  """randomised launch geometries of earl_tabletop_eval_episodes (batch sizes around the workgroup / group-count boundaries, episode lengths on and off
  the 8-step chunk grid, 2-40 episodes, every reset mode, both reward types, replayed or per-episode actions): every episode against the oracle"""
  import torch
  rng = np.random.default_rng(1000 + case)
  n = int(rng.choice([1, 63, 64, 65, 127, 1000, 2048, 4095, 4096, 4097, 8191, 8192, 8193, 12000]))
  T = int(rng.choice([16, 24, 32, 40, 56, 64, 72, 200] if rng.random() < 0.8 else [5, 17, 30]))       # (off-grid lengths take the launch-per-episode fallback)
  E = int(rng.integers(2, 41 if n * T < 400000 else 9))
  mode = int(rng.integers(0, 3))
  kw = dict(reward_type='dense' if rng.random() < 0.3 else 'sparse', wide_init=mode == 1, reset_at_goal=mode == 2, seed=int(rng.integers(0, 2**31)),
            env_offset=int(rng.integers(0, 1000)), horizon=T)
  o, h = orc.OracleTabletop(n, **kw), hx.HipTabletop(n, **kw)
  o.qpos[:] = rng.uniform(-2.8, 2.8, size=o.qpos.shape); o.attached[:] = rng.integers(-1, 1, size=n)
  o.steps_since_reset[:] = rng.integers(0, T, size=n); o.num_interventions[:] = rng.integers(0, 3, size=n)
  o.cfg.counter = int(rng.integers(0, 2**40))
  h.set_from(o)
  shared = rng.random() < 0.25
  g = torch.Generator(device='cuda').manual_seed(case)
  acts = (torch.rand(1 if shared else E, T, n, 3, generator=g, device='cuda') * 2 - 1).contiguous()
  acts[..., 2] = acts[..., 2].abs() * (torch.rand(acts.shape[:-1], generator=g, device='cuda') > 0.2)
  got = h.eval_episodes(acts[0], episodes=E) if shared else h.eval_episodes(acts)
  dense = kw['reward_type'] == 'dense'
  for e in range(E):
    o.reset()
    want = o.rollout(acts[0 if shared else e].cpu().numpy())
    assert_same_out(tuple(x[e].cpu().numpy() for x in got), want, dense)
  assert_same_state(o, h)
  assert h.cfg.counter == o.cfg.counter
