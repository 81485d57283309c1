"""Sawyer door env on the HIP stepper vs the CPU restatement (oracle/sawyer_oracle.py on oracle/physics_oracle.LinkModel),
and vs the reference's recorded demonstrations where those constrain it.

Dynamics parity with MuJoCo is UNPINNED (no simulator here); what the demos do pin before the gripper touches the handle:
the gripper opening (claw slide joints: actuator, armature, implicit damping, limits) to 1e-4, and loosely the hand path."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
LINKS = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'sawyer_door_links.npz')
DEMOS = os.path.join(REPO, 'earl_benchmark_amd', 'demonstrations', 'sawyer_door')


@pytest.fixture(scope='module')
def lm():
  from oracle import physics_oracle as po
  return po.LinkModel(LINKS)


def episodes(direction, k):
  z = np.load(os.path.join(DEMOS, direction, 'demo_data.npz'))
  t = np.nonzero(z['terminals'].ravel())[0]
  starts = [0] + list(t[:-1] + 1)
  return [(z['observations'][s], z['actions'][s:s + k], z['next_observations'][s:s + k]) for s in starts]


@pytest.mark.parametrize('reward_type,reset_at_goal', [('sparse', False), ('dense', True)])
def test_reset_and_rollout_match_oracle(lm, reward_type, reset_at_goal):
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from oracle.sawyer_oracle import SawyerDoorOracle
  n, T, off = 6, 7, 40
  env = PersistentStateWrapper(SawyerDoor(reward_type=reward_type, reset_at_goal=reset_at_goal, num_envs=n, seed=11, env_offset=off), 5)
  obs0 = env.reset().cpu().numpy()
  refs = [SawyerDoorOracle(lm, reward_type, reset_at_goal, seed=11, env_id=off + i, horizon=5) for i in range(n)]
  for r in refs:
    r._settled = refs[0].settle()
    r.counter = 1                       # the env constructor consumed draw 0
  ref0 = np.stack([r.reset() for r in refs])
  np.testing.assert_allclose(obs0, ref0, rtol=0, atol=1e-8)
  lo, hi = refs[0].angle_noise
  ang = env.unwrapped.qpos[:, 9].cpu().numpy() - refs[0].obj_init_angle
  assert (ang >= lo).all() and (ang <= hi).all() and len(np.unique(ang)) == n
  rng = np.random.default_rng(3)
  acts = rng.uniform(-1.3, 1.3, size=(T, n, 4)).astype(np.float32)
  out = env.rollout(torch.from_numpy(acts).cuda())
  for t in range(T):
    for i, r in enumerate(refs):
      o, rew, done, ok = r.step(acts[t, i])
      np.testing.assert_allclose(out['obs'][t, i].cpu().numpy(), o, rtol=0, atol=2e-8)
      assert abs(float(out['reward'][t, i]) - float(rew)) <= (0 if reward_type == 'sparse' else 1e-6)
      assert bool(out['done'][t, i]) == done and bool(out['success'][t, i]) == ok
  assert bool(out['done'][4].all()) and not bool(out['done'][3].any())
  # step() continues the same trajectory as rollout()
  o, rew, done, info = env.step(torch.from_numpy(acts[0]).cuda())
  ref = np.stack([r.step(acts[0, i])[0] for i, r in enumerate(refs)])
  np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=0, atol=5e-8)
  np.testing.assert_allclose(env._get_obs().cpu().numpy()[:, 7:], ref[:, 7:], atol=0)


def test_demo_prefixes_gripper_exact_hand_loose():
  """replay the first 12 actions of every demonstration episode from reset (before any contact)"""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  eps = episodes('forward', 12) + episodes('reverse', 12)
  n = len(eps)
  env = SawyerDoor(num_envs=n)
  obs0 = env.reset().cpu().numpy()
  # the reference's reset observation (sawyer_door.py:13; its own comment :45-47), gripper open, hand 5.9 / -0.3 / -5.1 mm off the mocap
  d0 = np.abs(obs0[:, :3] - eps[0][0][:3]).max(0)
  assert d0.max() < 4e-4 and (obs0[:, 3] == 1.0).all(), d0           # round 4: the recorded reset state (rounds 1 - 3: x off by 5.9 mm)
  acts = np.stack([e[1] for e in eps], axis=1)
  out = env.rollout(torch.from_numpy(acts).cuda())
  got = out['obs'].cpu().numpy()
  want = np.stack([e[2] for e in eps], axis=1)
  assert np.abs(got[..., 3] - want[..., 3]).max() < 2e-5                      # gripper opening (3e-6 measured)
  err = got[..., :3] - want[..., :3]
  assert np.sqrt((err ** 2).mean()) < 2e-3 and np.abs(err).max() < 5e-3       # hand path over the contact-free prefix (rounds 1 - 3: 8e-3 / 2.5e-2)
  nf = len(episodes('forward', 1))                                            # the reverse demos carry the reverse goal
  np.testing.assert_allclose(got[:, :nf, 7:], want[:, :nf, 7:], atol=1e-7)   # goal block (demos are float32)


def test_reward_and_success_on_demo_rows():
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from oracle.sawyer_oracle import compute_reward
  env_s, env_d = SawyerDoor(reward_type='sparse', num_envs=2), SawyerDoor(reward_type='dense', num_envs=2)
  for d in ('forward', 'reverse'):
    z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
    o = z['next_observations'].astype(np.float64)
    r = env_s.compute_reward(torch.from_numpy(o).cuda()).cpu().numpy()
    assert (r == z['rewards'].ravel()).all()                                    # the reference's recorded sparse rewards, bit-exact
    assert (env_s.is_successful(torch.from_numpy(o).cuda()).cpu().numpy() == (z['rewards'].ravel() == 1)).all()
    rd = env_d.compute_reward(torch.from_numpy(o).cuda()).cpu().numpy()
    hip = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
    want = np.array([compute_reward(row, 'dense', hip)[0] for row in o])
    np.testing.assert_allclose(rd, want, rtol=1e-6, atol=1e-6)


def test_info_dict_on_demo_rows_and_through_step():
  """the seven keys of SawyerDoorV2.evaluate_state (sawyer_door.py:127-139) -- every one a function of the observation -- against the
  restatement (oracle.sawyer_oracle.door_info) on all 1,095 demonstration rows, both reward types; NB 'success' is the 0.08 test, not is_successful()"""
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from oracle.sawyer_oracle import door_info
  hip = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
  for rt in ('sparse', 'dense'):
    env = SawyerDoor(reward_type=rt, num_envs=2)
    for d in ('forward', 'reverse'):
      z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
      o = z['next_observations'].astype(np.float64)
      got = {k: v.cpu().numpy() for k, v in env.info_from_obs(torch.from_numpy(o).cuda()).items()}
      assert set(got) == set(_abi.SAWYER_INFO_KEYS) == {'success', 'near_object', 'grasp_success', 'grasp_reward', 'in_place_reward', 'obj_to_target', 'unscaled_reward'}
      want = [door_info(row, rt, hip) for row in o]
      for k in got:
        np.testing.assert_allclose(got[k], np.array([w[k] for w in want]), rtol=1e-9, atol=1e-12, err_msg=f'{rt} {d} {k}')
      if rt == 'sparse':
        assert (got['unscaled_reward'] == z['rewards'].ravel()).all()            # = the recorded sparse reward
        assert (got['success'] >= z['rewards'].ravel()).all() and got['success'].sum() > z['rewards'].sum()   # the looser 0.08 radius
  # ... and the dict step() returns: the reference's keys + this build's own
  env = SawyerDoor(reward_type='sparse', num_envs=5, seed=3)
  env.reset()
  o, r, done, info = env.step(torch.zeros(5, 4))
  assert set(info) == set(_abi.SAWYER_INFO_KEYS) | {'is_successful', 'status'}
  want = [door_info(row, 'sparse', hip) for row in o.cpu().numpy()]
  for k in _abi.SAWYER_INFO_KEYS:
    np.testing.assert_allclose(info[k].cpu().numpy(), [w[k] for w in want], rtol=1e-9, atol=1e-12)
  one = SawyerDoor(reward_type='sparse', num_envs=1, seed=3, scalar_api=True)
  one.reset()
  _, _, _, i1 = one.step(np.zeros(4, np.float32))
  assert set(i1) == set(_abi.SAWYER_INFO_KEYS) and all(isinstance(v, float) for v in i1.values()) and i1['grasp_success'] == 1.0 and i1['near_object'] == 0.0


def test_info_dict_on_goal_switch_rows_uses_the_goal_the_reward_used():
  """with lifelong goal switching the kernel overwrites a switch row's goal block with the NEW goal, while the reference's evaluate_state -- and the
  row's reward -- ran before LifelongWrapper.reset_goal (lifelong_wrapper.py:30-44).  A custom goal (handle target right at the handle: success, dense reward 10)
  is replaced by the default goal at the first switch: on that row obs[7:] is the default goal already, but reward, info['unscaled_reward'],
  info['obj_to_target'] and info['success'] must still be those of the custom goal; through rollout() and through step()."""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.wrappers import LifelongWrapper, PersistentStateWrapper
  from oracle.sawyer_oracle import door_info
  hip = np.array([0, 0.4, 0.2], np.float32).astype(np.float64)
  n, T, gcf = 6, 7, 3
  for rt in ('sparse', 'dense'):
    for mode in ('rollout', 'step'):
      env = LifelongWrapper(PersistentStateWrapper(SawyerDoor(reward_type=rt, num_envs=n, seed=5), 1000), gcf)
      o0 = env.reset()
      custom = env.unwrapped.goal_t.clone()
      custom[:, 4:7] = o0[:, 4:7]                                         # target = where the handle is
      env.unwrapped.reset_goal(custom)
      default = torch.tensor(env.unwrapped.goal_states[0], dtype=torch.float64, device='cuda').expand(n, 7).contiguous()
      acts = torch.zeros(T, n, 4)
      if mode == 'rollout':
        out = env.rollout(acts)
        obs, rew, info = out['obs'], out['reward'], {k: out['info'][..., i] for i, k in enumerate(('success', 'near_object', 'grasp_success', 'grasp_reward',
                                                                                                    'in_place_reward', 'obj_to_target', 'unscaled_reward'))}
      else:
        rows = [env.step(acts[t]) for t in range(T)]
        obs, rew = torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])
        info = {k: torch.stack([r[3][k] for r in rows]) for k in ('success', 'obj_to_target', 'unscaled_reward', 'in_place_reward')}
      obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
      info = {k: v.cpu().numpy() for k, v in info.items()}
      sw = gcf - 1                                                          # the first switch row
      assert (obs[:sw, :, 7:] == custom.cpu().numpy()).all() and (obs[sw:, :, 7:] == default.cpu().numpy()).all()
      for t in range(T):
        np.testing.assert_array_equal(info['unscaled_reward'][t].astype(np.float32), rew[t], err_msg=f'{rt} {mode} row {t}')
        goal_used = (custom if t <= sw else default).cpu().numpy()
        for i in range(n):
          row = obs[t, i].copy(); row[7:] = goal_used[i]
          w = door_info(row, rt, hip)
          for k in info:
            np.testing.assert_allclose(info[k][t, i], w[k], rtol=1e-9, atol=1e-12, err_msg=f'{rt} {mode} row {t} {k}')
      assert (info['success'][:sw + 1] == 1.0).all() and (info['obj_to_target'][:sw + 1] < 1e-3).all()     # custom goal: on target, the switch row included
      assert (info['success'][sw + 1:] == 0.0).all() and (info['obj_to_target'][sw + 1:] > 0.1).all()      # default goal from the next row on


def test_a_collision_table_of_the_other_cone_is_refused_by_the_c_abi():
  """include/earl_physics.h: an entry point given a collision table compiled for the other friction cone returns EARL_ERR_ARG (until round 5 only
  the Python loader checked).  A copy of the door's table with the cone word flipped, handed to earl_sawyer_rollout directly."""
  import ctypes as C
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.physics import CollisionModelStruct
  env = SawyerDoor(num_envs=4, seed=1)
  env.reset()
  u = env
  cm = CollisionModelStruct.from_buffer_copy(bytes(u.model.col_struct))
  assert cm.cone == 1                                                       # the Sawyer scenes: elliptic (basic_scene.xml:2)
  cm.cone = 0
  raw = torch.frombuffer(bytearray(bytes(cm)), dtype=torch.uint8).cuda()
  out = u._new_out((1,))
  o = _abi.SawyerOut(obs=out['obs'].data_ptr(), reward=out['reward'].data_ptr(), done=out['done'].data_ptr(), success=out['success'].data_ptr(),
                     status=out['status'].data_ptr(), info=None)
  acts = torch.zeros(1, 4, 4, device='cuda')
  lib = _abi.load()
  rc = lib.earl_sawyer_rollout(u.model.buf.data_ptr(), raw.data_ptr(), u.nv, u._cfg_ref, u._st_ref, acts.data_ptr(), 1, C.byref(o), None)
  assert rc == -1                                                           # EARL_ERR_ARG
  rc = lib.earl_sawyer_rollout(u.model.buf.data_ptr(), u.model.col_ptr, u.nv, u._cfg_ref, u._st_ref, acts.data_ptr(), 1, C.byref(o), None)
  assert rc == 0
  torch.cuda.synchronize()
  # the cone word is remembered per device ADDRESS.  The same block rewritten with the right cone is still refused (stale entry) until its owner
  # announces the change -- earl_physics_forget_table -- and accepted afterwards; a freed table's entry must not outlive it either (DeviceModel does this on free)
  cm.cone = 1
  raw.copy_(torch.frombuffer(bytearray(bytes(cm)), dtype=torch.uint8))
  torch.cuda.synchronize()
  assert lib.earl_sawyer_rollout(u.model.buf.data_ptr(), raw.data_ptr(), u.nv, u._cfg_ref, u._st_ref, acts.data_ptr(), 1, C.byref(o), None) == -1
  assert lib.earl_physics_forget_table(raw.data_ptr()) == 1 and lib.earl_physics_forget_table(raw.data_ptr()) == 0
  assert lib.earl_sawyer_rollout(u.model.buf.data_ptr(), raw.data_ptr(), u.nv, u._cfg_ref, u._st_ref, acts.data_ptr(), 1, C.byref(o), None) == 0
  torch.cuda.synchronize()
  lib.earl_physics_forget_table(raw.data_ptr())


def test_envs_of_both_cones_built_and_dropped_in_turn_never_meet_a_stale_cone_entry():
  """every collision table is 55,648 B, so torch's caching allocator hands a freed Sawyer table's block (elliptic) to the next kitchen / minitaur table
  (pyramidal).  DeviceModel announces both ends of a table's life to the library; ten envs of alternating kinds in one process all run."""
  import gc
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.minitaur import Minitaur
  seen = set()
  for k in range(10):
    env = SawyerDoor(num_envs=4, seed=k) if k % 2 == 0 else Minitaur(num_envs=4, seed=k, scalar_api=False)
    seen.add((k % 2, env.model.col_ptr))
    env.reset()
    a = torch.zeros(2, 4, 4 if k % 2 == 0 else 8, device='cuda')
    env.rollout(a)
    torch.cuda.synchronize()
    del env, a
    gc.collect()
  assert len(seen) >= 2                                                 # (whether the allocator reuses a block across kinds is up to it: the loop must simply pass)


def test_loader_builds_the_door_env():
  import earl_benchmark_amd as eb
  import torch
  loader = eb.EARLEnvs('sawyer_door', reward_type='sparse', num_envs=4, eval_horizon=3)
  train, ev = loader.get_envs()
  o = ev.reset()
  assert o.shape == (4, 14) and o.dtype == torch.float64
  for t in range(3):
    o, r, done, info = ev.step(torch.zeros(4, 4))
  assert bool(done.all()) and int(ev.num_interventions[0]) == 1
  assert loader.get_initial_states().shape == (1, 7) and loader.get_goal_states().shape == (1, 7)
  np.testing.assert_allclose(o[:, 7:].cpu().numpy(), np.repeat(loader.get_goal_states(), 4, 0), atol=0)


def door_angle_for_handle(lm, oracle_env, handle_xyz):
  """invert the handle position recorded at the start of a demonstration episode -> door hinge angle"""
  angs = np.linspace(-1.5, 0.1, 3201)

  def handle(a):
    q = oracle_env.qpos.copy(); q[9] = a
    pos, quat, _ = lm.kinematics(q)
    return lm.attachment(pos, quat, oracle_env.k_obj)[0]
  return float(angs[int(np.argmin([((handle(a) - handle_xyz) ** 2).sum() for a in angs]))])


def test_contact_dynamics_match_oracle_through_a_grasp(lm):
  """forward demonstration 0 replayed from its recorded start: the gripper closes on the handle rod and drags the door.
  GPU and CPU statement are compared state by state (contacts, pyramidal friction, drag row, active-set Newton) over the
  steps where contacts are active, and the gripper opening is compared with what MuJoCo recorded."""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from oracle.sawyer_oracle import SawyerDoorOracle
  obs0, acts, nobs = episodes('forward', 44)[0]
  ref = SawyerDoorOracle(lm)
  ref.reset()
  ang = door_angle_for_handle(lm, ref, obs0[4:7].astype(np.float64))
  ref.qpos[9] = ang
  env = SawyerDoor(num_envs=3)
  env.reset()
  env.qpos[:, 9] = ang
  np.testing.assert_allclose(env.qpos[0].cpu().numpy(), ref.qpos, atol=1e-9)
  ncon_steps = 0
  for t in range(44):
    o_ref, r_ref, _, ok_ref = ref.step(acts[t])
    o, r, done, info = env.step(torch.from_numpy(np.tile(acts[t], (3, 1))).cuda())
    fwd = lm.forward(ref.qpos, ref.qvel, np.zeros(2), ref.mocap, np.array([1.0, 0, 1, 0]))
    ncon_steps += len(fwd['contacts']) > 0
    # contact problems are only piecewise smooth: a looser tolerance than the contact-free trajectories
    np.testing.assert_allclose(o[0].cpu().numpy(), o_ref, rtol=0, atol=1e-6, err_msg=f'step {t}')
    np.testing.assert_allclose(env.qpos[0].cpu().numpy(), ref.qpos, rtol=0, atol=1e-6, err_msg=f'step {t}')
    assert abs(o_ref[3] - nobs[t][3]) < 0.02, (t, o_ref[3], nobs[t][3])      # MuJoCo's recorded gripper opening, through the grasp
    # resynchronise so that every step is an independent comparison
    env.qpos[:] = torch.from_numpy(ref.qpos).cuda(); env.qvel[:] = torch.from_numpy(ref.qvel).cuda()
    env.mocap_pos[:] = torch.from_numpy(ref.mocap).cuda()
    assert np.linalg.norm(o_ref[4:7] - nobs[t][4:7]) < 0.006, (t, o_ref[4:7], nobs[t][4:7])   # ... and its recorded handle position
  assert ncon_steps >= 15
  assert ref.qpos[9] > ang + 0.15                      # the door was dragged towards closed
  assert bool((env.qpos[0] == env.qpos[1]).all())      # identical envs in one wavefront stay identical


def test_all_demo_episodes_open_loop_loose():
  """the 10 demonstration episodes replayed OPEN LOOP from their recorded start (door angle inverted from the first recorded handle position).  The
  demonstrations come from MuJoCo with a feedback policy; this build's stepper is a different simulator (handle rods as sphere chains + edge caps, one merged plate
  per finger, elliptic friction cone, 8-contact cap), with three declared calibrations.  The bounds below TRACK THE SHIPPED BUILD (VERDICT r04 item 1:
  never trail it): per-episode handle-path RMS as measured (profiles/r05_heldout_eval.json; the C restatement and the kernel agree to the tenth of a millimetre) plus
  a quarter, and the goal counts as measured:
  forward (close the door): 3.7 / 2.3 / 2.2 / 2.4 / 4.3 mm, three reach the goal (two end 9 mm short); gripper opening within 0.005, hand within 7 cm;
  reverse (pull the door open): 7.0 / 10.3 mm on the two short episodes, the three long ones lose the recorded contact sequence (30.9 / 54.5 / 59.3 mm); all five pull
  the door, one reaches the goal.
  Round 5 tried the reference's own contact geometry (two boxes per finger, box-cylinder narrow phase with one contact per pair, torsional rows) on the C restatement:
  forward unchanged within a millimetre, ALL FIVE reverse episodes lost (tests/test_contact_experiments.py) -- not shipped."""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  eps = []
  for d in ('forward', 'reverse'):
    z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
    ends = np.nonzero(z['terminals'].ravel())[0] + 1
    for s0, e0 in zip([0] + list(ends[:-1]), ends):
      eps.append((d, z['observations'][s0], z['actions'][s0:e0], z['next_observations'][s0:e0]))
  n, T = len(eps), max(len(e[2]) for e in eps)
  assert n == 10
  env = SawyerDoor(num_envs=n)
  env.reset()
  want_handle = np.stack([e[1][4:7] for e in eps]).astype(np.float64)
  best, err = np.zeros(n), np.full(n, 1e9)
  for a in np.linspace(-1.5, 0.1, 801):
    env.qpos[:, 9] = a
    e2 = ((env._get_obs()[:, 4:7].cpu().numpy() - want_handle) ** 2).sum(1)
    m = e2 < err
    best[m], err[m] = a, e2[m]
  assert np.sqrt(err.max()) < 1e-3
  env.qpos[:, 9] = torch.from_numpy(best).cuda()
  env.goal_t[:] = torch.from_numpy(np.stack([e[1][7:] for e in eps]).astype(np.float64)).cuda()
  acts = np.zeros((T, n, 4), np.float32)
  for i, e in enumerate(eps):
    acts[:len(e[2]), i] = e[2]
  out = env.rollout(torch.from_numpy(acts).cuda())
  obs, suc = out['obs'].cpu().numpy(), out['success'].cpu().numpy()
  assert np.isfinite(obs).all()
  reached = pulled = followed = rev_reached = 0
  SHIPPED_MM = {'forward': [3.7, 2.3, 2.2, 2.4, 4.3], 'reverse': [7.0, 10.3, 30.9, 54.5, 59.3]}      # profiles/r05_heldout_eval.json, episode order
  k = {'forward': 0, 'reverse': 0}
  for i, e in enumerate(eps):
    L = len(e[2])
    o, w = obs[:L, i], e[3]
    shipped = SHIPPED_MM[e[0]][k[e[0]]] * 1e-3
    k[e[0]] += 1
    handle_rms = np.sqrt(((o[:, 4:7] - w[:, 4:7]) ** 2).sum(1).mean())
    hand_max = np.linalg.norm(o[:, :3] - w[:, :3], axis=1).max()
    grip_max = np.abs(o[:, 3] - w[:, 3]).max()
    start = np.linalg.norm(w[0, 4:7] - w[0, 11:14])
    closest = np.linalg.norm(o[:, 4:7] - o[:, 11:14], axis=1).min()
    if e[0] == 'forward':
      reached += bool(suc[:L, i].any())
      assert handle_rms < 1.25 * shipped + 0.0005 and hand_max < 0.07 and grip_max < 0.005, (i, handle_rms, hand_max, grip_max)
    else:
      pulled += bool(closest < 0.95 * start)
      followed += bool(handle_rms < 0.02)
      rev_reached += bool(suc[:L, i].any())
      assert handle_rms < 1.25 * shipped + 0.0005, (i, handle_rms, closest, start)
    assert (o[:, 9 - 9 + 3] >= 0).all() and (np.abs(o[:, 6] - 0.10003595) < 1e-6).all()     # handle height never changes (hinge about z)
  assert reached >= 3 and pulled == 5 and followed >= 2 and rev_reached >= 1, (reached, pulled, followed, rev_reached)        # as shipped: 3, 5, 2, 1 (rounds 1 - 3: 4, 5, 3, 1)


def test_shards_equal_one_batch_and_both_lane_layouts_agree():
  """env-range sharding: two half shards (env_offset) reproduce one batch bit for bit, resets included (Philox streams are
  keyed by the global env id); and one-wavefront-per-env (64 lanes) gives the same numbers as four envs per wavefront"""
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  lib = _abi.load()
  n, T = 10, 12
  g = torch.Generator().manual_seed(5)
  acts = (torch.rand(T, n, 4, generator=g) * 2 - 1).cuda()
  acts[:, :, :3] = torch.tensor([0.9, 0.6, -0.9]).cuda()        # head for the table / door: contacts appear
  runs = {}
  try:
    for lanes in (16, 64):
      _abi.check(lib.earl_debug_set_physics_lanes(lanes), 'lanes')
      full = SawyerDoor(num_envs=n, seed=3)
      o0 = full.reset()
      out = full.rollout(acts)
      runs[lanes] = (o0.clone(), out['obs'].clone(), out['reward'].clone(), full.qpos.clone())
      if lanes == 16:
        a, b = SawyerDoor(num_envs=6, seed=3, env_offset=0), SawyerDoor(num_envs=4, seed=3, env_offset=6)
        oa, ob = a.reset(), b.reset()
        assert torch.equal(torch.cat([oa, ob]), o0)
        ra, rb = a.rollout(acts[:, :6].contiguous()), b.rollout(acts[:, 6:].contiguous())
        assert torch.equal(torch.cat([ra['obs'], rb['obs']], 1), out['obs'])
        assert torch.equal(torch.cat([a.qpos, b.qpos]), full.qpos)
  finally:
    _abi.check(lib.earl_debug_set_physics_lanes(16), 'lanes')
  for x, y in zip(runs[16], runs[64]):
    assert torch.equal(x, y)
  assert len(torch.unique(runs[16][0][:, 4])) == n              # every env drew its own door angle


def test_masked_reset_state_dict_and_lifelong_loader():
  import earl_benchmark_amd as eb
  import torch
  loader = eb.EARLEnvs('sawyer_door', reward_type='dense', setup_as_lifelong_learning=True, num_envs=5, train_horizon=7,
                       goal_change_frequency=3)
  env = loader.get_envs()
  o = env.reset()
  u = env.unwrapped
  acts = torch.zeros(4, 5, 4).cuda(); acts[..., 2] = -1.0
  out = env.rollout(acts)
  np.testing.assert_allclose(env.lifelong_return.cpu().numpy(), out['reward'].double().sum(0).cpu().numpy(), rtol=1e-12)
  sd = u.state_dict()
  before = u.qpos.clone()
  mask = torch.tensor([1, 0, 0, 1, 0], dtype=torch.bool).cuda()
  o2 = env.reset(mask=mask)
  assert torch.equal(u.qpos[~mask], before[~mask]) and not torch.equal(u.qpos[mask], before[mask])
  assert u.steps_since_reset.tolist() == [0, 4, 4, 0, 4] and env.num_interventions.tolist() == [2, 1, 1, 2, 1]
  np.testing.assert_allclose(o2[~mask].cpu().numpy(), u._get_obs()[~mask].cpu().numpy(), atol=0)
  u.load_state_dict(sd)
  assert torch.equal(u.qpos, before) and u.steps_since_reset.tolist() == [4] * 5
  o3, r3, d3, _ = env.step(acts[0])
  assert d3.tolist() == [False] * 5
  for _ in range(2):
    o3, r3, d3, _ = env.step(acts[0])
  assert d3.tolist() == [True] * 5                                # horizon 7 reached


def test_the_references_literal_reset_recipe_is_available():
  """reset_hand_timesteps=250: sim.reset() + 50 x (mocap at hand_init_pos, 5 timesteps) [UPSTREAM _reset_hand], the state the reference's episodes start
  from -- still moving.  The reference's own comment records its reset observation, hand = (0.00591636, 0.39968333, 0.19493164) (sawyer_door.py:45-47):
  5.9 mm off the mocap in x.  This stepper's 250-timestep state: 4.3 mm off in x, every coordinate within 3 mm of that constant (the converged
  default is 5.9 mm away in x); kernel == C restatement; the arm is not at rest (joint speeds of 0.5 rad/s)."""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from oracle import physics_c
  env = SawyerDoor(num_envs=3, seed=2, reset_hand_timesteps=250)
  obs = env.reset().cpu().numpy()
  rec = np.array([0.00591636, 0.39968333, 0.19493164])
  assert np.abs(obs[:, :3] - rec).max() < 3.2e-3 and (obs[:, 0] > 3e-3).all()
  assert 0.3 < float(env.qvel[:, :7].abs().max()) < 1.5
  cm = physics_c.CModel('sawyer_door')
  r = cm.run(np.zeros((1, cm.nv)), np.zeros((1, cm.nv)), np.array([0, 0.4, 0.2], np.float32).astype(float), [1, 0, 1, 0], [-1, 1], nsub=250)
  np.testing.assert_allclose(env.qpos[0, :9].cpu().numpy(), r['qpos'][0, :9], rtol=0, atol=1e-8)
  np.testing.assert_allclose(env.qvel[0, :9].cpu().numpy(), r['qvel'][0, :9], rtol=0, atol=1e-7)
  out = env.rollout(torch.zeros(5, 3, 4, device='cuda'))              # and it steps from there
  assert np.isfinite(out['obs'].cpu().numpy()).all() and int(out['status'].sum()) == 0
  conv = SawyerDoor(num_envs=1, seed=2, reset_state='converged')
  assert abs(float(conv.reset()[0]) - rec[0]) > 5e-3                # rounds 1 - 3's default: converged, x on the mocap
  dflt = SawyerDoor(num_envs=1, seed=2)                              # round 4's default: the state the recorded episodes start from (tables reset_*_recorded)
  assert dflt.reset_state == 'recorded' and np.abs(np.asarray(dflt.reset(), dtype=np.float64)[:3] - rec).max() < 4e-4
