"""BASELINE configs[2] at FULL size under pytest (sawyer_door and sawyer_peg, N = 8192 envs, the reference's eval horizons 300 / 200,
earl_benchmark/__init__.py:24-35), the per-env failure guard of the Sawyer kernels (include/earl_physics.h: earl_sawyer_out.status),
and the C-ABI corner cases the advisor listed (reset with obs == NULL, goal restored by reset).

Full-size checks are the size-independent properties the path offers -- every output finite and inside the arena, `done` exactly at
the horizon, two 4096-env shards == one 8192-env batch bit for bit -- plus a 64-env subset of the SAME batch compared with the CPU
restatement (oracle/physics_oracle.c) one env step at a time along the whole episode (the oracle is re-synchronised to the GPU state
before every step, so each of the 300 / 200 steps is an independent one-step comparison through whatever contacts that state has).
Dynamics parity with MuJoCo stays UNPINNED."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FULL = 8192
TASKS = {'sawyer_door': 300, 'sawyer_peg': 200}


def make(task, n, **kw):
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  cls = SawyerPeg if task == 'sawyer_peg' else SawyerDoor
  return PersistentStateWrapper(cls(num_envs=n, **kw), TASKS[task])


def actions(T, n, seed=5):
  import torch
  g = torch.Generator(device='cuda').manual_seed(seed)
  return (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).to(torch.float32)


@pytest.mark.parametrize('task', list(TASKS))
def test_full_size_episode_properties_shards_and_oracle_subset(task):
  import torch
  from oracle import physics_c
  T = TASKS[task]
  acts = actions(T, N_FULL)
  env = make(task, N_FULL, seed=21)
  obs0 = env.reset()
  out = env.rollout(acts)
  obs, rew, done, suc, status = (out[k] for k in ('obs', 'reward', 'done', 'success', 'status'))
  # --- properties at full size
  assert obs.shape == (T, N_FULL, 14) and bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all())
  assert bool(done[-1].all()) and not bool(done[:-1].any())                     # PersistentStateWrapper: first at step == horizon
  assert int(status.sum()) == 0 and int(env.unwrapped.fail_count.sum()) == 0    # no env diverged under random actions
  hand = obs[..., :3]
  lo, hi = torch.tensor([-0.55, 0.3, 0.0], device='cuda'), torch.tensor([0.55, 1.05, 0.55], device='cuda')
  assert bool((hand >= lo).all()) and bool((hand <= hi).all())                  # the hand follows the mocap, which is clipped to its box
  assert bool((obs[..., 3] >= 0).all()) and bool((obs[..., 3] <= 1).all())
  assert float(obs[..., 4:7].abs().max()) < 2.0                                 # the object stays on the table
  assert bool((obs[..., 7:] == obs0[None, :, 7:]).all())                        # goal block untouched
  assert bool(((rew == 1) == suc).all()) and bool(((rew == 0) | (rew == 1)).all())   # sparse reward == success flag
  u = env.unwrapped
  assert bool(torch.isfinite(u.qpos).all()) and bool(torch.isfinite(u.qvel).all()) and bool((u.steps_since_reset == T).all())
  np.testing.assert_array_equal(u.last_obs.cpu().numpy(), obs[-1].cpu().numpy())
  # --- two half shards == the batch, bit for bit (RNG keyed by the global env id)
  h = N_FULL // 2
  for k in range(2):
    sh = make(task, h, seed=21, env_offset=k * h)
    o0 = sh.reset()
    assert bool((o0 == obs0[k * h:(k + 1) * h]).all())
    so = sh.rollout(acts[:, k * h:(k + 1) * h].contiguous())
    for key in ('obs', 'reward', 'done', 'success', 'status'):
      assert bool((so[key] == out[key][:, k * h:(k + 1) * h]).all()), (task, k, key)
    del sh, so
  # --- 64 envs of the same batch against the CPU restatement, every step of the episode
  m = 64
  sub = make(task, m, seed=21)
  su = sub.unwrapped
  o0 = sub.reset()
  assert bool((o0 == obs0[:m]).all())
  cm = physics_c.CModel(task)
  cfg = (physics_c.peg_cfg if task == 'sawyer_peg' else physics_c.door_cfg)(att_names=cm.att_names, horizon=T)
  a_host = acts[:, :m].cpu().numpy()
  worst = 0.0
  for t in range(T):
    q, v, mp = su.qpos.cpu().numpy().copy(), su.qvel.cpu().numpy().copy(), su.mocap_pos.cpu().numpy().copy()
    goal, st = su.goal_t.cpu().numpy().copy(), su.steps_since_reset.cpu().numpy().copy()
    ob, r_ref, d_ref, ok_ref = cm.sawyer_rollout(cfg, q, v, mp, goal, st, a_host[t][None])
    o, r, d, info = sub.step(acts[t, :m])
    assert bool((o == obs[t, :m]).all())                                         # the subset IS the first 64 envs of the batch
    err = float(np.abs(o.cpu().numpy() - ob[0]).max())
    worst = max(worst, err, float(np.abs(su.qpos.cpu().numpy() - q).max()))
    assert err < 2e-6, (task, t, err)
    np.testing.assert_allclose(su.qvel.cpu().numpy(), v, rtol=0, atol=2e-4, err_msg=f'{task} step {t}')
    assert (r.cpu().numpy() == r_ref[0]).all() and (d.cpu().numpy() == d_ref[0]).all() and (info['is_successful'].cpu().numpy() == ok_ref[0]).all()
  assert worst < 2e-6, worst


@pytest.mark.parametrize('task', list(TASKS))
def test_failure_guard_rolls_back_one_env_and_leaves_its_neighbours_alone(task):
  """a poisoned env (NaN / runaway velocity written into its state row) is rolled back step after step: last stable observation, reward 0,
  status 1, fail_count; the other envs of the same wavefront (4 envs per wave) and batch are bit-identical to an unpoisoned run"""
  import torch
  from oracle import physics_c
  n, T = 12, 9
  acts = actions(T, n, seed=3)
  ref = make(task, n, seed=4)
  ref.reset()
  want = ref.rollout(acts)
  for poison, bad in ((float('nan'), 5), (1e3, 2)):
    env = make(task, n, seed=4)
    u = env.unwrapped
    obs0 = env.reset()
    u.qvel[bad, 1] = poison
    out = env.rollout(acts)
    ok = [i for i in range(n) if i != bad]
    for key in ('obs', 'reward', 'done', 'success', 'status'):
      assert bool((out[key][:, ok] == want[key][:, ok]).all()), (task, poison, key)     # neighbours untouched, bit for bit
    assert bool((u.qpos[ok] == ref.unwrapped.qpos[ok]).all()) and int(out['status'][:, ok].sum()) == 0
    st = out['status'][:, bad].cpu().numpy()
    assert st.sum() >= T - 1 and int(u.fail_count[bad]) == int(st.sum()) and int(u.fail_count.sum()) == int(st.sum())
    o = out['obs'][:, bad].cpu().numpy()
    assert np.isfinite(o).all()
    prev = obs0[bad].cpu().numpy()
    for t in range(T):
      if st[t]:
        np.testing.assert_array_equal(o[t], prev)                                      # the last stable observation, again
        assert float(out['reward'][t, bad]) == 0.0 and not bool(out['success'][t, bad])
      prev = o[t]
    assert bool(out['done'][-1, bad]) == bool(want['done'][-1, bad])                   # the rolled-back steps still count for the horizon
    if np.isnan(poison):
      # frozen from the first step on; identical to the CPU restatement's guard
      assert st.all() and bool(torch.isnan(u.qvel[bad, 1]))
      cm = physics_c.CModel(task)
      cfg = (physics_c.peg_cfg if task == 'sawyer_peg' else physics_c.door_cfg)(att_names=cm.att_names, horizon=TASKS[task])
      env2 = make(task, n, seed=4)
      u2 = env2.unwrapped
      ob0 = env2.reset().cpu().numpy()
      q, v, mp = u2.qpos.cpu().numpy().copy(), u2.qvel.cpu().numpy().copy(), u2.mocap_pos.cpu().numpy().copy()
      v[bad, 1] = poison
      status, last, fc = np.zeros((T, n), np.uint8), ob0.copy(), np.zeros(n, np.int32)
      ob, r_ref, d_ref, ok_ref = cm.sawyer_rollout(cfg, q, v, mp, u2.goal_t.cpu().numpy().copy(), np.zeros(n, np.int32), acts.cpu().numpy(),
                                                   last_obs=last, fail_count=fc, status=status)
      np.testing.assert_array_equal(status, out['status'].cpu().numpy())
      np.testing.assert_array_equal(fc, u.fail_count.cpu().numpy())
      np.testing.assert_array_equal(ob[:, bad], o)
      np.testing.assert_allclose(out['obs'].cpu().numpy(), ob, rtol=0, atol=1e-6)
    else:
      assert bool(torch.isfinite(u.qpos).all()) and bool(torch.isfinite(u.qvel).all())  # a finite start always leaves a finite state


def test_reset_without_obs_still_records_the_dense_reward_state_and_restores_the_goal():
  """C-ABI corner cases: earl_sawyer_reset(obs = NULL) must still write st.obj_init / st.last_obs (the peg's dense reward reads them);
  reset() puts the default goal back (reset_model -> reset_goal(), sawyer_door.py:123, sawyer_peg.py:195)"""
  import ctypes as C
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  env = SawyerPeg(reward_type='dense', num_envs=5, seed=9)
  want_obs = env.reset()
  want_init, counter = env.obj_init.clone(), env._cfg.counter
  env.obj_init.fill_(7.0)
  env.last_obs.fill_(7.0)
  env._cfg.counter = counter - 1                                  # replay the same draw, this time without an observation buffer
  rc = env._lib.earl_sawyer_reset(env.model.buf.data_ptr(), env.nv, C.byref(env._cfg), C.byref(env._st), env._reset_state[0].data_ptr(),
                                  env._reset_state[1].data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
  _abi.check(rc, 'earl_sawyer_reset')
  assert bool((env.obj_init == want_init).all()) and bool((env.last_obs == want_obs).all())
  for cls in (SawyerDoor, SawyerPeg):
    e = cls(num_envs=3)
    default = e.goal_t.clone()
    e.reset_goal(np.array([0.1, 0.5, 0.2, 1.0, 0.2, 0.6, 0.1]))
    assert not bool((e.goal_t == default).all(1).any()) and bool((e._get_obs()[:, 7:] == e.goal_t).all())
    o = e.reset(mask=torch.tensor([True, False, True]))
    assert bool((e.goal_t[0] == default[0]).all()) and bool((e.goal_t[2] == default[2]).all()) and not bool((e.goal_t[1] == default[1]).all())
    assert bool((o[:, 7:] == e.goal_t).all())


def test_stale_library_layout_is_refused(monkeypatch):
  from earl_benchmark_amd import _abi, physics
  lib = _abi.load()
  physics.check_layouts(lib)

  class Fake:
    earl_physics_model_size = staticmethod(lambda: 8)
    earl_physics_model24_size = lib.earl_physics_model24_size
    earl_collision_model_size = lib.earl_collision_model_size
    earl_sawyer_cfg_size = lib.earl_sawyer_cfg_size
    earl_minitaur_cfg_size = lib.earl_minitaur_cfg_size
  with pytest.raises(_abi.EarlHipError, match='sizeof'):
    physics.check_layouts(Fake)


def test_door_rollout_variants_are_bit_identical():
  """the door rollout kernel exists in two builds (csrc/physics.hip: four single-wave workgroups per CU; csrc/physics_w8.hip: one eight-wave
  workgroup per CU, packed matrices, in-LDS factorisations): same outputs and state, bit for bit, incl. a ragged last workgroup and contacts"""
  import torch
  from earl_benchmark_amd import _abi
  lib = _abi.load()
  n, T = 203, 60
  acts = actions(T, n, seed=8)
  acts[:, :, 1] = acts[:, :, 1].abs()           # drive the hands towards the door: contacts
  res = []
  try:
    for variant in (1, 2):
      assert lib.earl_debug_set_door_variant(variant) == 0
      env = make('sawyer_door', n, seed=2)
      env.reset()
      out = env.rollout(acts)
      res.append((out, env.unwrapped.qpos.clone(), env.unwrapped.qvel.clone()))
  finally:
    lib.earl_debug_set_door_variant(0)
  for key in ('obs', 'reward', 'done', 'success', 'status'):
    assert bool((res[0][0][key] == res[1][0][key]).all()), key
  assert bool((res[0][1] == res[1][1]).all()) and bool((res[0][2] == res[1][2]).all())
  assert lib.earl_debug_set_door_variant(4) != 0                 # 3 = the time-sliced schedule (measurement switch, tools/bench_door_schedule.py)


def test_peg_time_sliced_schedule_is_bit_identical_to_one_group_per_wave():
  """Round 4: for batches larger than one round the peg rollout is a queue of (env group, 10-step slice) items taken by persistent waves (csrc/physics.hip sched_claim;
  include/earl_physics.h earl_sawyer_state.sched).  An env's arithmetic does not depend on who runs it or when: every output and the final state equal the static
  schedule's bit for bit, for slice lengths that do and do not divide the rollout, with the lifelong goal switch on (its goal rows travel through HBM between slices)."""
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.wrappers import LifelongWrapper, PersistentStateWrapper
  lib = _abi.load()
  n, T = 8192, 23
  g = torch.Generator(device='cuda').manual_seed(31)
  acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
  acts[:, :, 2] = -acts[:, :, 2].abs()                                          # down to the table: contacts, grasps
  ref = None
  try:
    for k in (0, 1, 7, 2):                                                      # static; queue with slices of 10 (default), 7 and 2 env steps
      lib.earl_debug_set_peg_schedule(k)
      env = LifelongWrapper(PersistentStateWrapper(SawyerPeg(num_envs=n, seed=77, reset_at_goal=True), 100), 9)
      env.reset()
      out = env.rollout(acts)
      torch.cuda.synchronize()
      got = {kk: out[kk].clone() for kk in ('obs', 'reward', 'done', 'success', 'status', 'info')}
      got.update(qpos=env.unwrapped.qpos.clone(), qvel=env.unwrapped.qvel.clone(), goal=env.unwrapped.goal_t.clone(), steps=env.unwrapped.steps_since_reset.clone(),
                 last_obs=env.unwrapped.last_obs.clone(), lret=env.unwrapped.lifelong_return_t.clone())
      assert bool(torch.isfinite(got['obs']).all())
      if ref is None:
        ref = got
      else:
        for kk in ref:
          assert torch.equal(ref[kk], got[kk]), (k, kk)
    assert int((ref['goal'] != ref['goal'][0]).any()) == 1                       # goals were switched along the way (reverse task: 15 goal rows)
    sched = env.unwrapped.sched.cpu().numpy()
    G = (n + 3) // 4
    assert (sched[:G] == T).all() and (sched[G:] == 0).all()                     # every group ran to the end, every lock released
  finally:
    lib.earl_debug_set_peg_schedule(1)
