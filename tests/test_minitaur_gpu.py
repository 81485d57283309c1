"""Minitaur env on the HIP stepper (csrc/physics_mt.hip) against this build's CPU statement (oracle/physics_oracle.c via oracle.physics_c.CMinitaur,
itself checked against the numpy statement in tests/test_minitaur.py), through the C ABI / the env class.  PARITY WITH THE REFERENCE'S PYBULLET
SIMULATION IS UNPINNED AND MODEL-LESS: these tests pin the kernel to the CPU statement of the same model, nothing more."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope='module')
def torch():
  import torch
  return torch


def make(n, **kw):
  from earl_benchmark_amd.envs.minitaur import Minitaur
  return Minitaur(num_envs=n, scalar_api=False, **kw)


def test_reset_and_steps_match_the_cpu_statement(torch):
  from oracle import physics_c
  n, T = 37, 30
  env = make(n, seed=5, env_offset=3)
  c = physics_c.CMinitaur(n, seed=5, env_offset=3)
  c.reset()                                                    # (the constructor's reset)
  np.testing.assert_allclose(env.last_obs.cpu().numpy(), c.last_obs, rtol=0, atol=TOL)
  np.testing.assert_array_equal(env.goal_t.cpu().numpy(), c.goal)
  np.testing.assert_array_equal(env.motor_param.cpu().numpy(), c.motor_param)      # Philox draws: identical bits
  o = env.reset()
  oc = c.reset()
  np.testing.assert_allclose(o.cpu().numpy(), oc, rtol=0, atol=TOL)
  rng = np.random.default_rng(0)
  acts = rng.uniform(-1, 1, (T, n, 8)).astype(np.float32)
  worst = 0.0
  for t in range(T):                                           # per-step launches, re-synchronised per step (chaotic contacts: no long open-loop comparison)
    ob, r, done, info = env.step(torch.from_numpy(acts[t]).cuda())
    c.qpos[:], c.qvel[:] = c.qpos, c.qvel
    res = c.rollout(acts[t:t + 1])
    d = np.abs(ob.cpu().numpy() - res['obs'][0]).max()
    worst = max(worst, d)
    assert d < TOL, (t, d)
    np.testing.assert_allclose(r.cpu().numpy(), res['reward'][0], rtol=0, atol=TOL)
    np.testing.assert_array_equal(info['success'].cpu().numpy(), res['success'][0])
    # re-synchronise the CPU statement to the kernel's state
    c.qpos[:] = env.qpos.cpu().numpy(); c.qvel[:] = env.qvel.cpu().numpy()
    c.observed_torque[:] = env.observed_torque.cpu().numpy(); c.overheat[:] = env.overheat.cpu().numpy(); c.motor_enabled[:] = env.motor_enabled.cpu().numpy()
  assert not bool(done.any()) and int(env.fail_count.sum()) == 0
  z = env.qpos[:, 2]
  assert float(z.min()) > 0.05 and float(z.max()) < 0.3        # on the ground, not through it


def test_fused_rollout_equals_stepping_bit_for_bit_and_shards_equal_the_batch(torch):
  n, T = 48, 25
  a, b = make(n, seed=11), make(n, seed=11)
  g = torch.Generator(device='cuda').manual_seed(2)
  acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1)
  ra = a.rollout(acts)
  rows = [b.step(acts[t]) for t in range(T)]
  assert torch.equal(ra['obs'], torch.stack([r[0] for r in rows])) and torch.equal(ra['reward'], torch.stack([r[1] for r in rows]))
  assert torch.equal(ra['success'], torch.stack([r[3]['success'] for r in rows]))
  for k in ('qpos', 'qvel', 'overheat', 'motor_enabled', 'observed_torque', 'steps_since_reset'):
    assert torch.equal(getattr(a, k), getattr(b, k)), k
  # two half shards == one batch (Philox keyed by the global env id)
  s0, s1 = make(n // 2, seed=11), make(n // 2, seed=11, env_offset=n // 2)
  r0, r1 = s0.rollout(acts[:, :n // 2]), s1.rollout(acts[:, n // 2:])
  assert torch.equal(torch.cat([r0['obs'], r1['obs']], 1), ra['obs'])


def test_small_batch_launch_modes_are_bit_identical(torch):
  """Round 5: two envs per wave / one env per wave / one env per workgroup (include/earl_physics.h earl_debug_set_solo_mt; the launcher picks by batch size): same bits, reset
  (incl. the settle steps and the randomizer) and rollout, outputs and state"""
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  n, T = 45, 30
  g = torch.Generator(device='cuda').manual_seed(4)
  acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1)
  res = {}
  try:
    for mode in (0, 1, 2, -1):
      lib.earl_debug_set_solo_mt(mode)
      env = make(n, seed=13)
      out = env.rollout(acts)
      res[mode] = [out[k].clone() for k in ('obs', 'reward', 'done', 'success', 'status')] + [getattr(env, k).clone() for k in ('qpos', 'qvel', 'overheat', 'observed_torque')]
  finally:
    lib.earl_debug_set_solo_mt(-1)
  for mode in (1, 2, -1):
    for a, b in zip(res[0], res[mode]):
      assert torch.equal(a, b), mode


def test_tree_stepper_matches_the_generic_stepper(torch):
  """csrc/minitaur_stepper.h (arrow-shaped Hessian, legs eliminated before the root body, DPP exchanges) against the generic nv = 22 instantiation of
  csrc/physics.hip (dense factorisation): same algorithm, so the same numbers to rounding -- reset (100 settle timesteps) and env steps from
  identical states, re-synchronised per step (chaotic contacts amplify the last bits over long open loops)."""
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  n, T = 64, 40
  g = torch.Generator(device='cuda').manual_seed(5)
  acts = torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1
  try:
    lib.earl_debug_set_minitaur_stepper(0)
    ref = make(n, seed=21)
    o_ref = ref.reset().clone()
    lib.earl_debug_set_minitaur_stepper(1)
    new = make(n, seed=21)
    o_new = new.reset().clone()
    assert float((o_new - o_ref).abs().max()) < 1e-8, float((o_new - o_ref).abs().max())
    worst = 0.0
    for t in range(T):
      for k in ('qpos', 'qvel', 'overheat', 'motor_enabled', 'observed_torque'):
        getattr(ref, k).copy_(getattr(new, k))
      lib.earl_debug_set_minitaur_stepper(0)
      a = ref.step(acts[t])
      lib.earl_debug_set_minitaur_stepper(1)
      b = new.step(acts[t])
      torch.cuda.synchronize()
      d = max(float((a[0] - b[0]).abs().max()), float((ref.qpos - new.qpos).abs().max()), float((ref.qvel - new.qvel).abs().max()) * 1e-2)
      worst = max(worst, d)
      assert d < 1e-8, (t, d)
      assert torch.equal(a[3]['success'], b[3]['success']) and float((a[1] - b[1]).abs().max()) < 1e-8
  finally:
    lib.earl_debug_set_minitaur_stepper(1)
  assert int(new.fail_count.sum()) == 0


def test_loader_wrappers_goal_switch_and_action_bounds(torch):
  import earl_benchmark_amd as eb
  n = 16
  L = eb.EARLEnvs('minitaur', reward_type='dense', num_envs=n, seed=1, eval_horizon=4, allow_unpinned_dynamics=True)
  train, ev = L.get_envs()
  assert L.get_goal_states() is None and not L.has_demos()
  o = ev.reset()
  assert o.shape == (n, 32) and o.dtype == torch.float64
  with pytest.raises(ValueError, match='out of bounds'):
    ev.step(torch.full((n, 8), 1.5))
  nan_action = torch.zeros(n, 8)
  nan_action[3, 5] = float('nan')                                              # minitaur_gym_env.py:279: `not (lo <= x <= hi)` is True for NaN
  with pytest.raises(ValueError, match='5th action out of bounds'):
    ev.step(nan_action)
  for t in range(4):
    o, r, done, info = ev.step(torch.zeros(n, 8))
  assert bool(done.all()) and int(ev.num_interventions[0]) == 1 and ev.total_steps == 4
  assert torch.equal(ev.compute_reward(o), r)                                  # _reward == compute_reward on the returned observation
  assert torch.equal(ev.is_successful(o), info['success'])
  Ll = eb.EARLEnvs('minitaur', num_envs=n, seed=1, setup_as_lifelong_learning=True, goal_change_frequency=3, allow_unpinned_dynamics=True)
  lenv = Ll.get_envs()
  lenv.reset()
  g0 = lenv.unwrapped.goal_t.clone()
  res = lenv.rollout(torch.zeros(3, n, 8))
  assert torch.equal(res['obs'][1, :, 30:], g0) and torch.equal(res['obs'][2, :, 30:], lenv.unwrapped.goal_t)
  assert not torch.equal(lenv.unwrapped.goal_t, g0) and float(lenv.lifelong_return.abs().sum()) > 0


def test_failure_guard_rolls_back_one_env(torch):
  n = 8
  env = make(n, seed=4)
  ref = make(n, seed=4)
  env.qvel[3, 7] = float('nan')                                                # poison one env
  acts = torch.zeros(2, n, 8, device='cuda')
  ra, rb = env.rollout(acts), ref.rollout(acts)
  assert ra['status'][:, 3].tolist() == [1, 1] and int(ra['status'].sum()) == 2 and int(env.fail_count[3]) == 2
  keep = [i for i in range(n) if i != 3]
  assert torch.equal(ra['obs'][:, keep], rb['obs'][:, keep])                   # the neighbours are bit-identical to an unpoisoned run
  assert torch.equal(ra['obs'][0, 3], env.last_obs[3]) and float(ra['reward'][:, 3].abs().sum()) == 0.0


def test_full_size_soak_4096_envs_1000_steps(torch):
  """BASELINE configs[4] at size: 4096 envs, the reference's eval horizon (1000 env steps = 5000 timesteps), random actions"""
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  n, T = 4096, 1000
  env = PersistentStateWrapper(make(n, seed=77), T)
  g = torch.Generator(device='cuda').manual_seed(5)
  out = None
  for k in range(4):                                                           # four launches of 250 steps (obs rows: 262 MB each)
    acts = (torch.rand(T // 4, n, 8, generator=g, device='cuda') * 2 - 1)
    out = env.rollout(acts, out=out)
    assert bool(torch.isfinite(out['obs']).all())
  u = env.unwrapped
  assert bool(out['done'][-1].all()) and not bool(out['done'][:-1].any())
  assert int(u.fail_count.sum()) <= 4 and int((u.steps_since_reset == T).sum()) == n
  z = u.qpos[:, 2]
  assert float(z.min()) > 0.02 and float(z.max()) < 0.6                        # nobody fell through the ground or flew off
  assert float(u.qpos[:, :2].abs().max()) < 1.6                                # ... or through a wall


def test_the_env_randomizer_sets_masses_and_foot_friction_per_env(torch):
  """MinitaurEnvRandomizer [UPSTREAM] through the reference's own setters (minitaur.py:468-508): six parameters per env and reset (identical bits in the
  kernel and the CPU statement), each inside its range, different between envs and between resets; a heavier base sinks the stance lower; without
  the randomizer every env is the nominal model and the whole batch settles to ONE pose."""
  from oracle import physics_c
  n = 64
  env = make(n, seed=3)
  p0 = env.motor_param.cpu().numpy().copy()
  c = physics_c.CMinitaur(n, seed=3); c.reset()
  np.testing.assert_array_equal(p0, c.motor_param)
  assert ((p0[:, 0] >= 14.8) & (p0[:, 0] <= 16.8) & (p0[:, 1] >= 0) & (p0[:, 1] <= 0.01)).all()
  assert ((p0[:, 2] > 0.8) & (p0[:, 2] < 1.2)).all() and ((p0[:, 5] > 0.8) & (p0[:, 5] < 1.5)).all()
  up, lo = p0[:, 3] * 0.275, p0[:, 4] * 0.086                                      # masses of an upper / a lower link after SetLegMasses
  assert ((up > 0.8 * 0.275) & (up < 1.2 * 0.275)).all() and ((lo > 0.8 * 0.034) & (lo < 1.2 * 0.034)).all()
  assert all(len(np.unique(p0[:, k])) == n for k in range(6))
  z0 = env.qpos[:, 2].cpu().numpy().copy()
  assert np.corrcoef(p0[:, 2], z0)[0, 1] < -0.5                                    # heavier base -> lower stance after the 100 settle timesteps
  env.reset()
  p1 = env.motor_param.cpu().numpy()
  assert (p1 != p0).all()                                                          # new draws at every reset
  plain = make(n, seed=3, env_randomizer=None)
  pp = plain.motor_param.cpu().numpy()
  np.testing.assert_array_equal(pp, np.tile([16.0, 0.0, 1.0, 1.0, 1.0, -1.0], (n, 1)))
  assert float(plain.qpos.std(0).max()) < 1e-12
  only_motors = make(8, seed=3, env_randomizer=1)                                  # bit mask: 1 = voltage + damping only (what round 3's first minitaur build drew)
  pm = only_motors.motor_param.cpu().numpy()
  np.testing.assert_array_equal(pm[:, :2], p0[:8, :2]); np.testing.assert_array_equal(pm[:, 2:], np.tile([1.0, 1.0, 1.0, -1.0], (8, 1)))


def test_two_waves_per_simd_rollout_is_bit_identical_to_the_one_wave_kernel(torch):
  """Round 6: large batches take minitaur_duo_kernel (csrc/physics_env_minitaur.h: a timestep in two halves run by two waves of one SIMD, hand-over through LDS;
  picked by batch size, include/earl_physics.h earl_debug_set_minitaur_duo forces either).  Same expressions on the same values -- and, since the sums the two instantiations
  used to fuse differently are written as explicit fma chains (the orientation update of K10, the edge tests), THE SAME BITS: a batch and its shards, a fused rollout and its
  steps, agree whichever kernel each launch takes.  A batch that is no multiple of the workgroup's 16 envs, open loop over 40 env steps (200 timesteps), every output and the
  state; the kernel's fused rollout against its own stepping; the C statement on one step."""
  import numpy as np
  from earl_benchmark_amd import _abi
  from oracle import physics_c
  lib = _abi.load()
  n, T = 3083, 40                                                             # (3083 = 192 x 16 + 11: the last workgroup is ragged)
  g = torch.Generator(device='cuda').manual_seed(8)
  acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1)
  prev = lib.earl_debug_set_minitaur_duo(0)
  try:
    one = make(n, seed=31)
    ra = one.rollout(acts)
    lib.earl_debug_set_minitaur_duo(1)
    two, stepped = make(n, seed=31), make(n, seed=31)
    rb = two.rollout(acts)
    torch.cuda.synchronize()
    for k in ('obs', 'reward', 'done', 'success', 'status'):
      assert torch.equal(ra[k], rb[k]), k
    for k in ('qpos', 'qvel', 'overheat', 'motor_enabled', 'observed_torque', 'steps_since_reset', 'goal_t'):
      assert torch.equal(getattr(one, k), getattr(two, k)), k
    assert int(one.fail_count.sum()) == 0 and int(two.fail_count.sum()) == 0
    # fused rollout == stepping on the two-wave kernel (T = 1 launches take it too: same batch size)
    rows = [stepped.step(acts[t]) for t in range(8)]
    assert torch.equal(rb['obs'][:8], torch.stack([r[0] for r in rows])) and torch.equal(rb['reward'][:8], torch.stack([r[1] for r in rows]))
    # the launcher's own choice (by batch size) returns the same again
    lib.earl_debug_set_minitaur_duo(-1)
    auto = make(n, seed=31)
    assert torch.equal(auto.rollout(acts)['obs'], ra['obs'])
    # ... and the C statement, one env step from the reset state (the same call sequence on host arrays)
    lib.earl_debug_set_minitaur_duo(1)
    c = physics_c.CMinitaur(n, seed=31)
    c.reset()
    fresh = make(n, seed=31)
    assert float(np.abs(c.qpos - fresh.qpos.cpu().numpy()).max()) < 1e-8
    rc = c.rollout(acts[:1].cpu().numpy())
    rg = fresh.rollout(acts[:1])
    assert float(np.abs(rc['obs'][0] - rg['obs'][0].cpu().numpy()).max()) < 1e-6
  finally:
    lib.earl_debug_set_minitaur_duo(prev)


def test_two_waves_kernel_failure_guard_goal_switch_and_horizon_match_the_one_wave_kernel(torch):
  """the two-wave kernel's env-level paths against the one-wave kernel's, bit for bit, on a packed batch (1,100 envs; the kernel forced): a poisoned env is rolled back and flagged
  (its neighbours untouched), the lifelong wrapper's goal switch inside the launch (goal_change_frequency), the horizon's done flag and the wrappers' counters"""
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  n, T = 1100, 7
  g = torch.Generator(device='cuda').manual_seed(12)
  acts = (torch.rand(T, n, 8, generator=g, device='cuda') * 2 - 1)
  prev = lib.earl_debug_set_minitaur_duo(0)
  try:
    res = {}
    for mode in (0, 1):
      lib.earl_debug_set_minitaur_duo(mode)
      env = make(n, seed=41)
      env.qvel[5, 7] = float('nan')                                             # poison one env
      env._cfg.goal_change_frequency = 3                                        # LifelongWrapper.step's switch, every third env step
      env._cfg.horizon = 5
      out = env.rollout(acts)
      torch.cuda.synchronize()
      res[mode] = [out[k].clone() for k in ('obs', 'reward', 'done', 'success', 'status')] + \
                  [getattr(env, k).clone() for k in ('qpos', 'qvel', 'goal_t', 'fail_count', 'steps_since_reset', 'steps_since_goal_change', 'overheat', 'observed_torque', 'last_obs')]
    st = res[1][4]
    assert st[:, 5].tolist() == [1] * T and int(st.sum()) == T and int(res[1][8][5]) == T         # the poisoned env: every step rolled back and counted
    assert bool(res[1][2][4:].all()) and not bool(res[1][2][:4].any())                             # done from the horizon's step on
    for a, b in zip(res[0], res[1]):
      assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a.double()), torch.nan_to_num(b.double()))
  finally:
    lib.earl_debug_set_minitaur_duo(prev)
