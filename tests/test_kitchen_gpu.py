"""Kitchen env on the HIP stepper (BASELINE configs[3]) vs this build's CPU statement (oracle/physics_oracle.LinkModel,
oracle/kitchen_oracle.py): forward quantities and 40-timestep env steps incl. joint couplings, dry friction, springs, force limits and finger /
handle contacts; the loader surface; the reset recipe; a full-size (2048 envs x 400 steps) soak.
PARITY WITH MUJOCO IS UNPINNED for this env (the reference ships no recording of it and the simulator cannot run here)."""
import os

import numpy as np
import pytest

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu
LINKS = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'kitchen_links.npz')


@pytest.fixture(scope='module')
def lm():
  from oracle import physics_oracle as po
  return po.LinkModel(LINKS)


@pytest.fixture(scope='module')
def dm():
  from earl_benchmark_amd import physics
  return physics.DeviceModel('kitchen')


def T(a):
  import torch
  return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')


def test_layout_and_forward_match_the_cpu_statement(lm, dm):
  import ctypes as C
  from earl_benchmark_amd import physics
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  assert dm.lib.earl_physics_model24_size() == C.sizeof(physics.LinkModelStruct24) and dm.nv == 23 and dm.n_att == 10
  rng = np.random.default_rng(0)
  n = 8
  q = np.tile(INIT_QPOS, (n, 1)) + rng.normal(0, 0.05, (n, 23)); v = rng.normal(0, 0.3, (n, 23))
  q[:, 7:9] = rng.uniform(0, 0.04, (n, 2))
  q[1, 22] = -0.5; q[2, 19] = 0.3; q[3, 9] = -0.7; q[3, 10] = -0.004; q[4, 17] = -0.5; q[5, 20] = -1.7; q[6, 21] = 1.7    # open doors, couplings, limits
  v[7, 19:] = [3.0, -2.0, 2.0, -4.0]                                                                                    # sliding against dry friction
  mp = np.tile(MIDPOINT_POS, (n, 1)) + rng.normal(0, 0.05, (n, 3)); mq = np.tile(lm.weld_mocap_quat, (n, 1)); ctrl = rng.uniform(-0.01, 0.05, (n, 2))
  qacc, efc, att = dm.forward(T(q), T(v), T(mp), T(mq), T(ctrl))
  for i in range(n):
    r = lm.forward(q[i], v[i], ctrl[i], mp[i], mq[i])
    np.testing.assert_allclose(qacc[i].cpu().numpy(), r['qacc'], rtol=1e-9, atol=1e-8 * np.abs(r['qacc']).max(), err_msg=f'env {i}')
    a_ref = np.stack([lm.attachment(r['pos'], r['quat'], k)[0] for k in range(dm.n_att)])
    np.testing.assert_allclose(att[i].cpu().numpy(), a_ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(efc[i, :6].cpu().numpy(), r['f'][:6], rtol=1e-7, atol=1e-7 * np.abs(r['f'][:6]).max())   # the six weld forces


def test_env_steps_match_the_cpu_statement_through_a_grasp(lm):
  """the env (glue + 40 timesteps + noise-free observation + reward) against oracle/kitchen_oracle.py, resynchronised every env step: first free
  motion with random actions, then the hand is driven onto the microwave handle with closing fingers (sphere-chain / finger-box contacts)."""
  import torch
  from earl_benchmark_amd.envs.kitchen import Kitchen
  from oracle import glue_oracle as go
  from oracle.kitchen_oracle import KitchenOracle
  g = load_golden('kitchen_step')
  ref = KitchenOracle(go.kitchen_params(g['kitchen_pos_bound'], g['kitchen_vel_bound'], g['kitchen_pos_noise_amp']), lm)
  env = Kitchen(num_envs=3, sensor_noise=False, seed=2)
  obs = env.reset()
  assert obs.shape == (3, 46) and bool(torch.isfinite(obs).all())
  rng = np.random.default_rng(1)
  names = [str(x) for x in lm.att_names]
  ncon_steps, worst = 0, 0.0
  for t in range(26):
    if t < 6:
      a = rng.uniform(-1, 1, 9).astype(np.float32)
    else:                                       # steer the mocap towards a point in front of the microwave handle, then close the fingers
      pos, quat, _ = lm.kinematics(env.qpos[0].cpu().numpy())
      handle = lm.attachment(pos, quat, names.index('microhandle_site'))[0]
      ee = lm.attachment(pos, quat, names.index('end_effector'))[0]
      d = (handle - ee) / 0.02
      a = np.concatenate([np.clip(d, -1, 1), np.zeros(4), [-1.0, -1.0] if t > 16 else [1.0, 1.0]]).astype(np.float32)
    ref.set(env.qpos[0].cpu().numpy(), env.qvel[0].cpu().numpy(), env.mocap_pos[0].cpu().numpy(), env.goal_t[0].cpu().numpy(), env.last_qp_robot[0].cpu().numpy())
    o_ref, r_ref, s_ref, out = ref.step(a)
    o, r, done, info = env.step(torch.from_numpy(np.tile(a, (3, 1))).cuda())
    ncon_steps += len(out['contacts']) > 0
    err = max(np.abs(o[0].cpu().numpy() - o_ref).max(), np.abs(env.qpos[0].cpu().numpy() - ref.qpos).max())
    worst = max(worst, err)
    assert err < 1e-6, (t, err)
    np.testing.assert_allclose(env.qvel[0].cpu().numpy(), ref.qvel, rtol=0, atol=1e-5, err_msg=f'step {t}')
    assert abs(float(r[0]) - r_ref) < 1e-6 * max(1.0, abs(r_ref)) and bool(info['success'][0]) == s_ref
    np.testing.assert_allclose(env.mocap_pos[0].cpu().numpy(), ref.mocap, atol=0)
  assert ncon_steps >= 3, ncon_steps                    # the fingers did touch the handle chain
  assert int(env.fail_count.sum()) == 0 and worst < 1e-6


def test_loader_reset_recipe_noise_and_wrappers():
  import torch
  import earl_benchmark_amd as eb
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS
  with pytest.raises(ValueError, match='only supports dense'):
    eb.EARLEnvs('kitchen', reward_type='sparse')
  n = 64
  L = eb.EARLEnvs('kitchen', reward_type='dense', num_envs=n, seed=5, eval_horizon=3, allow_unpinned_dynamics=True)
  train, ev = L.get_envs()
  assert L.get_initial_states().shape == (6, 23) and L.get_goal_states().shape == (1, 23) and not L.has_demos()
  u = ev.unwrapped
  o = ev.reset()
  assert o.shape == (n, 46) and o.dtype == torch.float64
  init = torch.tensor(L.get_initial_states(), dtype=torch.float64, device='cuda')
  # reset_model (kitchen.py:118-139): the fixtures start at one of the six all_pairs rows (every row occurs in a batch of 64); the goal is goal_states[0]
  d = (u.qpos[:, None, 9:] - init[None, :, 9:]).abs().amax(2)
  assert float(d.amin(1).max()) < 0.05 and len(torch.unique(d.argmin(1))) == 6
  assert bool((o[:, 23:] == torch.tensor(L.get_goal_states()[0], device='cuda')).all())
  # sensor noise (franka_robot.py:137-168): |obs - qpos| <= 0.1 * amp, and it differs between envs and between calls
  amp = torch.tensor(np.ctypeslib.as_array(u._params.pos_noise_amp), device='cuda')
  assert bool(((o[:, :23] - u.qpos).abs() <= 0.1 * amp + 1e-15).all()) and float((o[:, :9] - u.qpos[:, :9]).abs().max()) > 1e-4
  o2 = ev._get_obs()
  assert not bool((o2[:, :9] == o[:, :9]).all())
  # the arm was pulled towards the mocap midpoint for 400 timesteps; the finger targets are the clamped joints 0 / 1 (0.04, 0)
  assert float((u.qpos[:, :7] - torch.tensor(INIT_QPOS[:7], device='cuda')).abs().max()) > 0.05 and float((u.qpos[:, 7] - 0.04).abs().max()) < 5e-3
  for t in range(3):
    o, r, done, info = ev.step(torch.zeros(n, 9))
  assert bool(done.all()) and int(ev.num_interventions[0]) == 1 and ev.total_steps == 3 and r.dtype == torch.float64
  # shards == batch (Philox keyed by the global env id)
  from earl_benchmark_amd.envs.kitchen import Kitchen
  a = Kitchen(num_envs=16, seed=9); b0 = Kitchen(num_envs=8, seed=9); b1 = Kitchen(num_envs=8, seed=9, env_offset=8)
  act = (torch.rand(16, 9, device='cuda') * 2 - 1)
  oa = a.step(act)[0]; ob = torch.cat([b0.step(act[:8])[0], b1.step(act[8:])[0]])
  assert bool((oa == ob).all())


@pytest.mark.parametrize('task', ['microwave', 'light_switch', 'slide_cabinet', 'hinge_cabinet', 'light_slide'])
def test_reset_of_a_named_task_moves_only_that_tasks_fixtures(task):
  """Kitchen(task=...).reset_model (kitchen.py:122-126): `reset_pos[9:] = initial_states[task][9:]` -- the one row of the named task, not a
  draw over the 'all_pairs' table ('microwave' used to reset to micro_hinge).  Checked on the fixture joints after the settle."""
  import torch
  import earl_benchmark_amd as eb
  from earl_benchmark_amd import tables
  L = eb.EARLEnvs('kitchen', reward_type='dense', num_envs=8, seed=3, kitchen_task=task, allow_unpinned_dynamics=True)
  _, ev = L.get_envs()
  u = ev.unwrapped
  ev.reset()
  row = torch.tensor(tables.get(f'kitchen_task_{task}')[0], device='cuda')
  goal = torch.tensor(tables.goal_states('kitchen')[0], device='cuda')
  moved = (row[9:] - goal[9:]).abs() > 1e-3                                       # the fixtures this task displaces from the clean goal state
  assert int(moved.sum()) in (1, 2, 3, 4)
  d = (u.qpos[:, 9:] - row[None, 9:]).abs()
  assert float(d.max()) < 0.05                                                   # every env at the task's row (400 settle timesteps move fixtures by < 0.05)
  assert float((u.qpos[:, 9:][:, ~moved] - goal[None, 9:][:, ~moved]).abs().max()) < 0.05   # ... and the other fixtures at their goal positions
  with pytest.raises(KeyError, match='the reference defines'):
    eb.EARLEnvs('kitchen', reward_type='dense', num_envs=2, kitchen_task='burner0', allow_unpinned_dynamics=True).get_envs()


def test_the_hand_stops_at_the_counter_top_and_at_the_hood():
  """Round 3: the hand (link 7: flange, hand hull, finger envelope as eight spheres) against the kitchen's big static boxes
  (tools/mjcf_compile.py kitchen: counter-top slab, oven / stove body, back wall, hood, microwave body, cabinet bottoms; likewise the wrist, the forearm
  and the finger boxes' corner points: 56 blocks, the kitchen kernel's near masks are 64 bits wide).  The mocap target is driven
  0.1 m INTO the counter top (from above) and 0.05 m into the hood's front face (from the front; 0.1 m there gives 8 mm) through the raw stepper -- the env's own clip box,
  kitchen_multitask_v0.py:49-50, keeps the target above z = 1.8 and in front of y = 0.5 --: the weld pulls, the hand stops at the surface
  (penetration < 5 mm at the counter, < 8 mm at the hood), and the kernel equals the CPU statement through those contacts."""
  import torch
  from earl_benchmark_amd import physics
  from earl_benchmark_amd.envs.kitchen import INIT_QPOS, MIDPOINT_POS
  from oracle import physics_c, physics_oracle as po
  dm = physics.DeviceModel('kitchen')
  cm = physics_c.CModel('kitchen')
  tb = dm.tables
  hand = [i for i in range(len(tb['col_sph_link'])) if tb['col_sph_link'][i] == 6 and tb['col_sph_r'][i] > 0.015]
  assert len(hand) == 8 and len(tb['col_blk_begin']) == 61 and len(tb['col_box_link']) == 13      # (round 4: + the right-hand counter, five blocks)
  lm = po.LinkModel({k: tb[k] for k in tb})
  kw = dict(dtype=torch.float64, device='cuda')
  mq = torch.tensor(tb['weld_mocap_quat'], **kw)[None].contiguous()
  ctrl = torch.tensor([[0.04, 0.0]], **kw)
  rad = tb['col_sph_r'][hand]

  def centres(q):
    pos, quat, _ = lm.kinematics(q)
    return np.array([pos[6] + po.quat_mat(quat[6]) @ tb['col_sph_pos'][i] for i in hand])
  # boxes 6 and 9 of the table: the counter-top slab (top face z = 1.60), the hood (front face y = 0.675, z in [2.164, 2.404])
  slab, hood = (tb['col_box_pos'][6], tb['col_box_half'][6]), (tb['col_box_pos'][9], tb['col_box_half'][9])
  assert abs(slab[0][2] + slab[1][2] - 1.60) < 1e-6 and abs(hood[0][1] - hood[1][1] - 0.675) < 1e-3
  # box 12 (round 4): the right-hand counter, hull of its body and the slab pieces around the sink (top face z = 1.60, x from 0.498)
  rc = (tb['col_box_pos'][12], tb['col_box_half'][12])
  assert abs(rc[0][2] + rc[1][2] - 1.60) < 1e-6 and abs(rc[0][0] - rc[1][0] - 0.498) < 2e-3
  cases = (('counter top', lambda f: (-0.7, 0.1, 2.226 + (1.50 - 2.226) * f),
            lambda c: 1.60 - (c[:, 2] - rad).min()),
           ('right counter', lambda f: (-0.2 + (0.58 + 0.2) * min(1.0, 2 * f), 0.1 + (0.30 - 0.1) * min(1.0, 2 * f), 2.226 + (1.50 - 2.226) * max(0.0, 2 * f - 1)),   # over it, then down
                                                                                                         # (beyond the env's clip box x <= 0.4: the raw stepper)
            None),      # (at that reach the arm is stretched and tilted: whichever of forearm / wrist / hand / finger corners is lowest over the counter touches)
           ('hood front', lambda f: (-0.2, 0.1 + (0.725 - 0.1) * f, 2.226 + (2.28 - 2.226) * f),
            lambda c: ((c[:, 1] + rad)[(c[:, 2] > hood[0][2] - hood[1][2]) & (c[:, 2] < hood[0][2] + hood[1][2])]).max() - 0.675))
  for name, target, penetration in cases:
    q, v = torch.tensor(INIT_QPOS, **kw)[None].contiguous(), torch.zeros(1, 23, **kw)
    dm.step(q, v, torch.tensor([MIDPOINT_POS], **kw), mq, ctrl, nsub=400)
    worst = 0.0
    for step in range(80):
      mp = torch.tensor([target(min(1.0, step / 60.0))], **kw)
      qc, vc = q.cpu().numpy().copy(), v.cpu().numpy().copy()
      dm.step(q, v, mp, mq, ctrl, nsub=40)
      if step >= 60:                                    # in contact: one-env-step comparisons with the C restatement (re-synchronised per step)
        r = cm.run(qc, vc, mp.cpu().numpy()[0], tb['weld_mocap_quat'], [0.04, 0.0], nsub=40)
        assert r['ncon'][0] >= 1
        worst = max(worst, float(np.abs(r['qpos'] - q.cpu().numpy()).max()))
    if penetration is None:
      pos, quat, _ = lm.kinematics(q.cpu().numpy()[0])
      arm = [i for i in range(len(tb['col_sph_link'])) if tb['col_sph_link'][i] >= 4 and tb['col_sph_link'][i] <= 8]     # forearm, wrist, hand spheres + the finger boxes' corner points
      c = np.array([pos[tb['col_sph_link'][i]] + po.quat_mat(quat[tb['col_sph_link'][i]]) @ tb['col_sph_pos'][i] for i in arm])
      over = (np.abs(c[:, 0] - rc[0][0]) < rc[1][0]) & (np.abs(c[:, 1] - rc[0][1]) < rc[1][1])
      pen = float(1.60 - (c[over, 2] - tb['col_sph_r'][arm][over]).min())
    else:
      pen = float(penetration(centres(q.cpu().numpy()[0])))
    assert 0.0 < pen < (8e-3 if name == 'hood front' else 5e-3), (name, pen)   # resting AT the surface: in contact; 5-10 cm of weld pull give millimetres of
                                                        # soft-constraint penetration (counter 3.6 mm, hood 7 mm: MuJoCo-style impedance rows, not rigid stops)
    assert worst < 1e-6, (name, worst)
    assert bool(torch.isfinite(q).all())


def test_full_size_soak_2048_envs_400_steps():
  """BASELINE configs[3] at size: 2048 envs, the reference's eval horizon (400 env steps = 16,000 timesteps), random actions"""
  import torch
  from earl_benchmark_amd.envs.kitchen import Kitchen
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  n, T = 2048, 400
  env = PersistentStateWrapper(Kitchen(num_envs=n, seed=3), T)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(0)
  u = env.unwrapped
  moved = torch.zeros(n, 14, dtype=torch.float64, device='cuda')
  q0 = u.qpos[:, 9:].clone()
  for t in range(T):
    a = torch.rand(n, 9, generator=g, device='cuda') * 2 - 1
    o, r, done, info = env.step(a)
    moved = torch.maximum(moved, (u.qpos[:, 9:] - q0).abs())
    if t % 50 == 0 or t == T - 1:
      assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all()), t
  assert bool(done.all()) and int(info['status'].sum()) == 0 and int(u.fail_count.sum()) == 0
  # joint limits are SOFT rows (MuJoCo's solref / solimp): a finger driven by the stiff mocap weld into a knob (inertia 1e-3 kg m^2) pushes
  # it 0.2-0.5 rad past its stop while the contact lasts (r = torque (1 - d) invweight / (k d^2) = 0.018 rad per N m); the arm's own
  # limits give 0.1-0.45 rad under the weld's pull.  Bounded, and the fixtures do get operated by random actions:
  lo = torch.tensor(u.model.tables['jnt_range'][:, 0], device='cuda') - 0.8
  hi = torch.tensor(u.model.tables['jnt_range'][:, 1], device='cuda') + 0.8
  assert bool(((u.qpos >= lo) & (u.qpos <= hi)).all())
  assert float(moved[:, [0, 2, 4, 6]].max()) > 0.5 and float(moved[:, 12].max()) > 0.3     # some knob was turned, some hinge door opened
  f64 = dict(dtype=torch.float64, device='cuda')
  assert float((u.mocap_pos - torch.tensor([[-0.7, -0.1, 1.8]], **f64)).min()) >= 0 and float((u.mocap_pos - torch.tensor([[0.4, 0.5, 2.6]], **f64)).max()) <= 0
  assert float(u.qvel.abs().max()) < 50


def test_fused_rollout_equals_stepping_bit_for_bit():
  """earl_kitchen_rollout (ONE launch, T env steps) == T calls of earl_kitchen_step: outputs and every piece of state, incl. the sensor noise
  (same Philox counters), through fixture contacts (200 steps of random actions reach them), for a batch that does not fill its last workgroup"""
  import torch
  from earl_benchmark_amd.envs.kitchen import Kitchen
  n, T = 203, 200
  a_env, b_env = Kitchen(num_envs=n, seed=11), Kitchen(num_envs=n, seed=11)
  g = torch.Generator(device='cuda').manual_seed(5)
  acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1
  acts[:, :40, 2] -= 0.6                                   # some hands go down to the counter's knobs and stay in contact
  acts[3, 7] = float('nan')                                # a NaN action: the guard trips in env 7 at step 3 in both paths
  a_env.reset(); b_env.reset()
  b_env._fused_step = False                                  # the per-step C entry point (earl_kitchen_step, eight launches) is the other side
  fused = a_env.rollout(acts)
  rows = [b_env.step(acts[t]) for t in range(T)]
  assert torch.equal(fused['obs'].view(torch.int64), torch.stack([r[0] for r in rows]).view(torch.int64))
  assert torch.equal(fused['reward'].view(torch.int64), torch.stack([r[1] for r in rows]).view(torch.int64))
  assert torch.equal(fused['done'], torch.stack([r[2] for r in rows])) and torch.equal(fused['success'], torch.stack([r[3]['success'] for r in rows]))
  assert torch.equal(fused['status'], torch.stack([r[3]['status'] for r in rows]))
  assert int(fused['status'][3, 7]) == 1 and int(fused['status'].sum()) >= 1 and int(a_env.fail_count[7]) >= 1
  # the diverged step is rolled back INCLUDING its mocap target (it used to stay NaN and pull every later step of the env into the guard):
  # env 7 carries on from its last stable state at step 4, its attachment positions never hold the NaN the stepper left
  assert int(fused['status'][:, 7].sum()) == 1 and int(a_env.fail_count[7]) == 1 and bool(torch.isfinite(a_env.mocap_pos).all())
  assert bool(torch.isfinite(a_env.att).all()) and torch.equal(a_env.att.view(torch.int64), b_env.att.view(torch.int64))
  assert bool(torch.isfinite(a_env.compute_reward(a_env.last_obs)).all())
  for k in ('qpos', 'qvel', 'mocap_pos', 'last_qp_robot', 'last_obs', 'steps_since_reset', 'fail_count'):
    x, y = getattr(a_env, k), getattr(b_env, k)
    assert torch.equal(x.view(torch.int64) if x.dtype == torch.float64 else x, y.view(torch.int64) if y.dtype == torch.float64 else y), k
  ok = (fused['status'][-1] == 0)
  assert torch.equal(a_env.att[ok].view(torch.int64), b_env.att[ok].view(torch.int64))
  assert a_env._counter == b_env._counter and a_env.total_step_count == b_env.total_step_count
  # and the next rollout continues from there (counters, cached readings)
  more = torch.rand(3, n, 9, generator=g, device='cuda') * 2 - 1
  f2 = a_env.rollout(more)
  r2 = [b_env.step(more[t]) for t in range(3)]
  assert torch.equal(f2['obs'].view(torch.int64), torch.stack([r[0] for r in r2]).view(torch.int64))
  # step() itself goes through the fused kernel with T = 1 by default: the same again
  c_env = Kitchen(num_envs=n, seed=11)
  c_env.reset()
  for t in range(12):
    o, r, d, info = c_env.step(acts[t])
    assert torch.equal(o.view(torch.int64), fused['obs'][t].view(torch.int64)) and torch.equal(r.view(torch.int64), fused['reward'][t].view(torch.int64)), t
    assert torch.equal(info['status'], fused['status'][t])


@pytest.mark.parametrize('n', [37, 300])
def test_small_batch_launch_modes_are_bit_identical(n):
  """Round 5: the kitchen launch gives every env of a small batch a wave (n <= 4 x CUs) or a whole workgroup (n <= CUs: one wave, or all four waves on the
  env's timestep -- mode 3, the default there) to itself instead of packing two envs into a wave (include/earl_physics.h earl_debug_set_solo).  Every mode -- and so every pairing of envs in a wave -- returns the same bits, outputs and state, through fixture
  contacts: an env's result does not depend on which env shares its wave (the solver's coupled path gives an untouched env the bits of the uncoupled one)."""
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.kitchen import Kitchen
  lib = _abi.load()
  T = 60
  g = torch.Generator(device='cuda').manual_seed(9)
  acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1
  acts[:, ::2, 2] -= 0.6                                   # every other hand goes down to the knobs: wave-mates on different solver paths in the packed mode
  res = {}
  try:
    for mode in (0, 1, 2, 3, 4, -1):
      assert lib.earl_debug_set_solo(mode) in (-1, 0, 1, 2, 3, 4)
      env = Kitchen(num_envs=n, seed=21)
      env.reset()
      out = env.rollout(acts)
      res[mode] = ({k: v.clone() for k, v in out.items()}, env.qpos.clone(), env.qvel.clone(), env.att.clone())
  finally:
    lib.earl_debug_set_solo(-1)
  ref = res[0]
  touched = int((ref[0]['obs'][:, :, 9:23] - ref[0]['obs'][0, :, 9:23]).abs().amax(0).amax(1).gt(1e-3).sum())
  assert touched >= 3, touched                                 # fixtures were moved: contacts happened
  for mode in (1, 2, 3, 4, -1):
    for k in ref[0]:
      a, b = ref[0][k], res[mode][0][k]
      assert torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b), (mode, k)
    for a, b in zip(ref[1:], res[mode][1:]):
      assert torch.equal(a.view(torch.int64), b.view(torch.int64)), mode


def test_fused_rollout_full_size():
  """2048 envs x 400 steps in one launch: finite, done exactly at the horizon, nobody diverges; the first 64 envs equal a 64-env batch stepped one step at a time"""
  import torch
  from earl_benchmark_amd.envs.kitchen import Kitchen
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  n, T = 2048, 400
  env = PersistentStateWrapper(Kitchen(num_envs=n, seed=3), T)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(0)
  acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1
  out = env.unwrapped.rollout(acts)
  assert bool(torch.isfinite(out['obs']).all()) and bool(torch.isfinite(out['reward']).all())
  assert bool(out['done'][-1].all()) and not bool(out['done'][:-1].any()) and int(out['status'].sum()) == 0
  small = PersistentStateWrapper(Kitchen(num_envs=64, seed=3), T)
  small.reset()
  for t in range(60):
    o, r, d, info = small.step(acts[t, :64])
    assert torch.equal(o.view(torch.int64), out['obs'][t, :64].view(torch.int64)), t
    assert torch.equal(r.view(torch.int64), out['reward'][t, :64].view(torch.int64)), t


def test_scalar_api_lifelong_wrapper_and_state_dict():
  """the reference's call surface for ONE env (numpy obs, python float reward, bool done, {}), the LifelongWrapper goal switch, checkpointing"""
  import torch
  import earl_benchmark_amd as eb
  L = eb.EARLEnvs('kitchen', reward_type='dense', eval_horizon=2, allow_unpinned_dynamics=True)
  train, ev = L.get_envs()
  o = ev.reset()
  assert isinstance(o, np.ndarray) and o.shape == (46,) and o.dtype == np.float64
  o, r, d, info = ev.step(np.zeros(9))
  assert isinstance(r, float) and d is False and set(info) == {'time', 'obs_dict', 'rewards', 'score', 'images'} and isinstance(ev.is_successful(), bool)
  o, r, d, info = ev.step(np.zeros(9))
  assert d is True and ev.total_steps == 2 and ev.num_interventions == 1
  assert abs(ev.compute_reward(o) - r) < 1e-9                                   # the reward of the CURRENT simulator state, like the reference's
  LL = eb.EARLEnvs('kitchen', reward_type='dense', setup_as_lifelong_learning=True, num_envs=4, goal_change_frequency=3, allow_unpinned_dynamics=True)
  env = LL.get_envs()
  env.reset()
  u = env.unwrapped
  custom = torch.tensor(L.get_goal_states()[0], device='cuda').clone(); custom[22] = -1.0
  u.reset_goal(custom.cpu().numpy())
  rets = torch.zeros(4, dtype=torch.float64, device='cuda')
  for t in range(3):
    o, r, d, info = env.step(torch.zeros(4, 9))
    rets += r
    if t < 2:
      assert bool((o[:, 23:] == custom).all())
  assert bool((o[:, 23:] == torch.tensor(L.get_goal_states()[0], device='cuda')).all())      # switched back by reset_goal() on the third step: NEW goal in the obs
  assert float(r[0]) < 0                                                                       # ... while the reward of that step used the OLD one (microwave far open)
  np.testing.assert_allclose(env.lifelong_return.cpu().numpy(), rets.cpu().numpy())
  sd = u.state_dict()
  a = torch.rand(4, 9, device='cuda') * 2 - 1
  o1 = env.step(a)[0].clone()
  u.load_state_dict(sd)
  assert bool((env.step(a)[0] == o1).all())                                                    # same state + same Philox counter -> same step, noise included


def test_env_info_dict_of_the_reference_step():
  """KitchenV0.step's env_info -- 'time', 'obs_dict' (t, qp, qv, obj_qp, obj_qv, goal), 'rewards' (true_reward = r_total),
  'score', 'images' (adept_envs/franka/kitchen_multitask_v0.py:116-123, envs/kitchen.py:141-175) -- plus this build's own keys.  The velocity readings carry
  draws 9-17 / 32-45 of the step's 46 uniforms (franka_robot.py:155-159) times 0.1 x vel_noise_amp (franka_config.xml)."""
  import torch
  from earl_benchmark_amd.envs.kitchen import FRAME_SKIP, STREAM_NOISE, VEL_NOISE_AMP, Kitchen
  n = 6
  env = Kitchen(num_envs=n, seed=11)
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(1)
  for k in range(3):
    a = torch.rand(n, 9, generator=g, device='cuda') * 2 - 1
    counter = env._counter
    o, r, done, info = env.step(a)
    assert set(info) == {'time', 'obs_dict', 'rewards', 'score', 'images', 'success', 'is_successful', 'status'}
    od = info['obs_dict']
    assert set(od) == {'t', 'qp', 'qv', 'obj_qp', 'obj_qv', 'goal'}
    np.testing.assert_allclose(info['time'].cpu().numpy(), (10 + k + 1) * FRAME_SKIP * 0.002, rtol=0, atol=1e-12)     # sim.reset() + ten robot steps at the reset
    assert torch.equal(od['t'], info['time']) and torch.equal(od['qp'], o[:, :9]) and torch.equal(od['obj_qp'], o[:, 9:23]) and torch.equal(od['goal'], o[:, 23:])
    assert torch.equal(info['rewards']['r_total'], r) and torch.equal(info['rewards']['true_reward'], r) and float(info['score'].abs().max()) == 0.0 and info['images'] == []
    # the velocity readings: the state's velocities + the SAME draws the observation's positions used (stream / counter of this step)
    env._counter, keep = counter, env._counter
    u = env._uniform(46, STREAM_NOISE, -1.0, 1.0)
    env._counter = keep
    amp = torch.as_tensor(0.1 * VEL_NOISE_AMP, device='cuda')
    want = env.qvel + amp * torch.cat([u[:, 9:18], u[:, 32:46]], 1)
    assert torch.equal(torch.cat([od['qv'], od['obj_qv']], 1), want)
    noise = (torch.cat([od['qv'], od['obj_qv']], 1) - env.qvel).abs()
    assert bool((noise <= amp + 1e-15).all()) and float(noise.max()) > 0.0
    pos_noise = (o[:, :23] - env.qpos).abs()                                      # (and the positions': draws 0-8 / 18-31)
    assert float(pos_noise.max()) > 0.0
  quiet = Kitchen(num_envs=2, seed=11, sensor_noise=False)
  quiet.reset()
  _, _, _, i2 = quiet.step(torch.zeros(2, 9))
  assert torch.equal(torch.cat([i2['obs_dict']['qv'], i2['obs_dict']['obj_qv']], 1), quiet.qvel)
  one = Kitchen(num_envs=1, seed=11, scalar_api=True)
  one.reset()
  _, r1, _, i1 = one.step(np.zeros(9, np.float32))
  assert set(i1) == {'time', 'obs_dict', 'rewards', 'score', 'images'} and i1['rewards']['r_total'] == r1 and i1['score'] == 0.0
  assert abs(i1['time'] - 11 * FRAME_SKIP * 0.002) < 1e-12 and i1['obs_dict']['qv'].shape == (9,) and i1['obs_dict']['obj_qv'].shape == (14,)
