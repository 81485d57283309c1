"""Sawyer peg-insertion env on the HIP stepper vs the CPU restatement (oracle/sawyer_oracle.SawyerPegOracle on
oracle/physics_oracle.LinkModel, and the C restatement for the long runs), and vs the reference's recorded demonstrations
where those constrain it.

Dynamics parity with MuJoCo is UNPINNED (no simulator here).  Pinned by the reference's data: the sparse rule (bit-exact on
all 1,815 demonstration rows), the reset observation (hand within 8 mm, gripper 1.0, pegHead z = 0.02 and xy inside the
reset box), the reverse-task reset (peg in the hole, goal = one of the 15 initial states).  Loose: the 10 forward
demonstrations replayed open loop (grasp, lift, insert)."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
LINKS = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'sawyer_peg_links.npz')
DEMOS = os.path.join(REPO, 'earl_benchmark_amd', 'demonstrations', 'sawyer_peg')


@pytest.fixture(scope='module')
def lm():
  from oracle import physics_oracle as po
  return po.LinkModel(LINKS)


def episodes(direction):
  z = np.load(os.path.join(DEMOS, direction, 'demo_data.npz'))
  ends = np.nonzero(z['terminals'].ravel())[0] + 1
  return [(z['observations'][s], z['actions'][s:e], z['next_observations'][s:e], z['rewards'][s:e].ravel())
          for s, e in zip([0] + list(ends[:-1]), ends)]


@pytest.mark.parametrize('reset_at_goal', [False, True])
def test_reset_and_rollout_match_oracle(lm, reset_at_goal):
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.wrappers import PersistentStateWrapper
  from oracle.sawyer_oracle import PEG_INITIAL_STATES, SawyerPegOracle
  n, T, off = 6, 7, 40
  env = PersistentStateWrapper(SawyerPeg(reset_at_goal=reset_at_goal, num_envs=n, seed=11, env_offset=off), 5)
  obs0 = env.reset().cpu().numpy()
  refs = [SawyerPegOracle(lm, 'sparse', reset_at_goal, seed=11, env_id=off + i, horizon=5) for i in range(n)]
  for r in refs:
    r._settled = refs[0].settle()
    r.counter = 1                       # the env constructor consumed draw 0
  ref0 = np.stack([r.reset() for r in refs])
  np.testing.assert_allclose(obs0, ref0, rtol=0, atol=1e-8)
  u = env.unwrapped
  np.testing.assert_allclose(u.qpos.cpu().numpy(), np.stack([r.qpos for r in refs]), atol=1e-9)
  assert (u.qvel[:, 9:] == 0).all()
  if reset_at_goal:                     # peg in the hole, goal drawn from the initial-state table (sawyer_peg.py:149-152, :216-227)
    assert (np.abs(obs0[:, 4:7] - [-0.27, 0.6, 0.13]) <= 0.02 + 1e-12).all()
    assert all((np.abs(PEG_INITIAL_STATES - g).max(1) == 0).any() for g in obs0[:, 7:])
    assert len(np.unique(obs0[:, 11])) > 1
  else:
    assert (obs0[:, 4] >= -0.1).all() and (obs0[:, 4] <= 0.1).all() and (obs0[:, 5] >= 0.5).all() and (obs0[:, 5] <= 0.7).all()
    np.testing.assert_allclose(obs0[:, 6], 0.02, atol=1e-5)      # _set_obj_xyz keeps the settled orientation: 0.1 m x a 1.5e-5 rad pitch
    assert len(np.unique(obs0[:, 4])) == n
  rng = np.random.default_rng(3)
  acts = rng.uniform(-1.3, 1.3, size=(T, n, 4)).astype(np.float32)
  acts[:, :, 2] = -np.abs(acts[:, :, 2])            # head down, towards the peg / the table
  out = env.rollout(torch.from_numpy(acts).cuda())
  for t in range(T):
    for i, r in enumerate(refs):
      o, rew, done, ok = r.step(acts[t, i])
      np.testing.assert_allclose(out['obs'][t, i].cpu().numpy(), o, rtol=0, atol=1e-7)
      assert float(out['reward'][t, i]) == float(rew)
      assert bool(out['done'][t, i]) == done and bool(out['success'][t, i]) == ok
  assert bool(out['done'][4].all()) and not bool(out['done'][3].any())
  o, rew, done, info = env.step(torch.from_numpy(acts[0]).cuda())
  ref = np.stack([r.step(acts[0, i])[0] for i, r in enumerate(refs)])
  np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=0, atol=2e-7)


def test_reset_observation_against_the_demonstrations():
  """every recorded episode starts with the reference's own reset observation"""
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg, goal_states, initial_states
  fwd, rev = episodes('forward'), episodes('reverse')
  env = SawyerPeg(num_envs=16, seed=1)
  o = env.reset().cpu().numpy()
  first = np.stack([e[0] for e in fwd]).astype(np.float64)
  assert np.abs(o[:, :3] - first[0, :3]).max() < 5e-4            # hand at reset: the recorded start state since round 4 (rounds 1 - 3, converged pose: within 8 mm)
  assert np.abs(o[:, 3] - 1.0).max() < 1e-9 and (first[:, 3] == 1.0).all()
  np.testing.assert_allclose(o[:, 7:], np.repeat(goal_states, 16, 0), atol=0)
  np.testing.assert_allclose(first[:, 7:], np.repeat(goal_states, len(first), 0), atol=1e-7)
  assert (np.abs(first[:, 6] - 0.02) < 1e-7).all() and (np.abs(o[:, 6] - 0.02) < 1e-5).all()
  for x in (o, first):                                           # pegHead = peg - (0.1, 0, 0), peg ~ U([0, 0.2] x [0.5, 0.7])
    assert (x[:, 4] >= -0.1 - 1e-7).all() and (x[:, 4] <= 0.1).all() and (x[:, 5] >= 0.5).all() and (x[:, 5] <= 0.7).all()
  env_r = SawyerPeg(num_envs=16, seed=1, reset_at_goal=True)
  o_r = env_r.reset().cpu().numpy()
  first_r = np.stack([e[0] for e in rev]).astype(np.float64)
  for x, tol in ((o_r, 1e-12), (first_r, 1e-6)):                 # the reverse task starts in the hole, its goal is an initial state
    assert (np.abs(x[:, 4:7] - [-0.27, 0.6, 0.13]) <= 0.02 + tol).all()
    assert all((np.abs(initial_states - g).max(1) < 1e-6).any() for g in x[:, 7:])


def test_sparse_rule_on_demo_rows():
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  env = SawyerPeg(num_envs=2)
  rows = 0
  for d in ('forward', 'reverse'):
    z = np.load(os.path.join(DEMOS, d, 'demo_data.npz'))
    o = torch.from_numpy(z['next_observations'].astype(np.float64)).cuda()
    assert (env.compute_reward(o).cpu().numpy() == z['rewards'].ravel()).all()     # the reference's recorded sparse rewards, bit-exact
    assert (env.is_successful(o).cpu().numpy() == (z['rewards'].ravel() == 1)).all()
    rows += len(z['rewards'])
  assert rows == 1815
  with pytest.raises(NotImplementedError):           # the dense reward reads simulator state: step / rollout only
    SawyerPeg(reward_type='dense', num_envs=2).compute_reward(o)


def place_pegs(env, heads):
  import torch
  env.qpos[:, 9:12] = torch.from_numpy(heads + np.array([0.1, 0.0, 0.0])).cuda()
  env.qvel[:, 9:] = 0


def test_grasp_and_lift_match_the_cpu_statement():
  """forward demonstration 5 (grasp, lift, carry, insert) from its recorded start: HIP env vs the C restatement of the same
  algorithm, resynchronised every step so that each env step is an independent comparison through plate / peg / table /
  hole-block contacts; and the outcome of the open-loop replay itself."""
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from oracle import physics_c
  obs0, acts, nobs, rew = episodes('forward')[5]
  T = len(acts)
  env = SawyerPeg(num_envs=3)
  env.reset()
  place_pegs(env, np.tile(obs0[4:7].astype(np.float64), (3, 1)))
  cm = physics_c.CModel('sawyer_peg')
  cfg = physics_c.peg_cfg(att_names=cm.att_names)
  q, v, mp = env.qpos[:1].cpu().numpy().copy(), env.qvel[:1].cpu().numpy().copy(), env.mocap_pos[:1].cpu().numpy().copy()
  goal, st = env.goal_t[:1].cpu().numpy().copy(), np.zeros(1, np.int32)
  lifted, ncon_steps = 0.0, 0
  for t in range(T):
    ob, r_ref, _, ok_ref = cm.sawyer_rollout(cfg, q, v, mp, goal, st, acts[t][None, None, :])
    o, r, done, info = env.step(torch.from_numpy(np.tile(acts[t], (3, 1))).cuda())
    np.testing.assert_allclose(o[0].cpu().numpy(), ob[0, 0], rtol=0, atol=1e-6, err_msg=f'step {t}')
    np.testing.assert_allclose(env.qpos[0].cpu().numpy(), q[0], rtol=0, atol=1e-6, err_msg=f'step {t}')
    assert float(r[0]) == float(r_ref[0, 0]) and bool(info['is_successful'][0]) == bool(ok_ref[0, 0])
    ncon_steps += int(cm.run(q, v, mp, [1.0, 0, 1, 0], [0.0, 0.0], integrate=False)['ncon'][0] > 2)
    lifted = max(lifted, ob[0, 0, 6])
    env.qpos[:] = torch.from_numpy(q).cuda(); env.qvel[:] = torch.from_numpy(v).cuda(); env.mocap_pos[:] = torch.from_numpy(mp).cuda()
  assert ncon_steps >= 15                              # the plates held the peg for a good part of the episode
  assert lifted > 0.12 and abs(lifted - nobs[:, 6].max()) < 0.02      # lifted as high as MuJoCo's recording (13.0 cm)
  assert bool(info['is_successful'][0]) and rew[-1] == 1.0   # ... and inserted: the open-loop replay ends in the hole, like the demonstration
  assert bool((env.qpos[0] == env.qpos[1]).all())      # identical envs in one wavefront stay identical


def test_forward_demos_open_loop_loose():
  """the 10 forward demonstrations (MuJoCo, feedback policy) replayed OPEN LOOP in this build's stepper (sphere-chain
  peg, elliptic friction cone since round 4, 12-contact cap).  Only loose agreement is asserted; the bounds are what this round measures plus
  margin (DESIGN.md quotes the measured values with the calibrated weld: hand RMS 0.6-0.9 cm, peg RMS 0.4-1.5 cm, 10 / 10 lifted to the
  recorded height, 7 / 10 inserted): the hand follows the recorded path (RMS < 1.2 cm), so does the peg (RMS < 1.8 cm), at least 9 episodes
  lift the peg to within 2 cm of the recorded height, at least 5 end inserted.
  Round 4 (weld and start state identified on the contact-free prefixes): hand RMS 0.3 - 0.6 cm, peg RMS 0.2 - 0.8 cm -- with the pyramidal cone of that intermediate build 2 - 3 of 10 inserted; the
  shipped elliptic cone inserts 7 of 10, and the bound below tracks that."""
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  eps = episodes('forward')
  n, T = len(eps), max(len(e[1]) for e in eps)
  assert n == 10
  env = SawyerPeg(num_envs=n)
  env.reset()
  place_pegs(env, np.stack([e[0][4:7] for e in eps]).astype(np.float64))
  acts = np.zeros((T, n, 4), np.float32)
  for i, e in enumerate(eps):
    acts[:len(e[1]), i] = e[1]
  out = env.rollout(torch.from_numpy(acts).cuda())
  obs, suc = out['obs'].cpu().numpy(), out['success'].cpu().numpy()
  assert np.isfinite(obs).all()
  lifted = inserted = 0
  for i, e in enumerate(eps):
    L = len(e[1])
    o, w = obs[:L, i], e[2]
    assert np.sqrt(((o[:, :3] - w[:, :3]) ** 2).sum(1).mean()) < 0.007, i            # rounds 1 - 3: 0.012
    assert np.sqrt(((o[:, 4:7] - w[:, 4:7]) ** 2).sum(1).mean()) < 0.010, i          # rounds 1 - 3: 0.018
    assert (o[:, 6] > 0.004).all()                      # the peg is pressed into the soft table top by the plates at most ~1 cm, never through it
    lifted += abs(o[:, 6].max() - w[:, 6].max()) < 0.02
    inserted += bool(suc[L - 1, i])
  assert lifted >= 9 and inserted >= 6, (lifted, inserted)                          # shipped (elliptic cone): 10 lifted, 7 inserted


def test_shards_equal_one_batch_and_both_lane_layouts_agree():
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  lib = _abi.load()
  n, T = 10, 10
  g = torch.Generator().manual_seed(5)
  acts = (torch.rand(T, n, 4, generator=g) * 2 - 1).cuda()
  acts[:, :, 2] = -0.9                                           # down onto the table / the pegs
  runs = {}
  try:
    for lanes in (16, 64):
      _abi.check(lib.earl_debug_set_physics_lanes(lanes), 'lanes')
      full = SawyerPeg(num_envs=n, seed=3)
      o0 = full.reset()
      out = full.rollout(acts)
      runs[lanes] = (o0.clone(), out['obs'].clone(), out['reward'].clone(), full.qpos.clone())
      if lanes == 16:
        a, b = SawyerPeg(num_envs=6, seed=3, env_offset=0), SawyerPeg(num_envs=4, seed=3, env_offset=6)
        oa, ob = a.reset(), b.reset()
        assert torch.equal(torch.cat([oa, ob]), o0)
        ra, rb = a.rollout(acts[:, :6].contiguous()), b.rollout(acts[:, 6:].contiguous())
        assert torch.equal(torch.cat([ra['obs'], rb['obs']], 1), out['obs'])
        assert torch.equal(torch.cat([a.qpos, b.qpos]), full.qpos)
  finally:
    _abi.check(lib.earl_debug_set_physics_lanes(16), 'lanes')
  for x, y in zip(runs[16], runs[64]):
    assert torch.equal(x, y)


def test_loader_masked_reset_and_lifelong():
  import earl_benchmark_amd as eb
  import torch
  loader = eb.EARLEnvs('sawyer_peg', reward_type='sparse', num_envs=5, eval_horizon=3)
  train, ev = loader.get_envs()
  o = ev.reset()
  assert o.shape == (5, 14) and o.dtype == torch.float64
  for t in range(3):
    o, r, done, info = ev.step(torch.zeros(5, 4))
  assert bool(done.all()) and int(ev.num_interventions[0]) == 1
  assert loader.get_initial_states().shape == (15, 7) and loader.get_goal_states().shape == (1, 7)
  np.testing.assert_allclose(o[:, 7:].cpu().numpy(), np.repeat(loader.get_goal_states(), 5, 0), atol=0)
  u = ev.unwrapped
  before = u.qpos.clone()
  mask = torch.tensor([1, 0, 0, 1, 0], dtype=torch.bool).cuda()
  ev.reset(mask=mask)
  assert torch.equal(u.qpos[~mask], before[~mask]) and not torch.equal(u.qpos[mask], before[mask])
  assert u.steps_since_reset.tolist() == [0, 3, 3, 0, 3]
  life = eb.EARLEnvs('sawyer_peg', reward_type='sparse', setup_as_lifelong_learning=True, reset_train_env_at_goal=True, num_envs=4,
                     train_horizon=6, goal_change_frequency=2).get_envs()
  o = life.reset()
  out = life.rollout(torch.zeros(3, 4, 4).cuda())
  assert out['obs'].shape == (3, 4, 14)


def test_wide_init_reset_matches_oracle(lm):
  """wide_init (sawyer_peg.py:200-209): half of the resets use the default reset box, the other half a row of the wide table + noise"""
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg, wide_initial_states
  from oracle.sawyer_oracle import SawyerPegOracle
  n = 48
  env = SawyerPeg(num_envs=n, seed=21, wide_init=True)
  obs0 = env.reset().cpu().numpy()
  refs = [SawyerPegOracle(lm, seed=21, env_id=i, wide_init=True) for i in range(n)]
  for r in refs:
    r._settled = refs[0].settle()
    r.counter = 1
  ref0 = np.stack([r.reset() for r in refs])
  np.testing.assert_allclose(obs0, ref0, rtol=0, atol=1e-8)
  peg = obs0[:, 4:7] + [0.1, 0, 0]
  default = (peg[:, 0] >= 0) & (peg[:, 0] <= 0.2) & (peg[:, 1] >= 0.5) & (peg[:, 1] <= 0.7) & (np.abs(peg[:, 2] - 0.02) < 1e-5)
  near_table = np.array([np.abs(wide_initial_states + [0.1, 0, 0] - p).max(1).min() <= 0.02 + 1e-9 for p in peg])
  assert (default | near_table).all() and 12 <= default.sum() <= 36 and 12 <= near_table.sum() <= 36


def test_lifelong_goal_switch_matches_oracle(lm):
  """LifelongWrapper on the reverse task (reset_at_goal: goals are the 15 initial states): every goal_change_frequency steps the goal
  is redrawn, the returned observation carries the NEW goal, the reward of that step was computed with the old one"""
  import earl_benchmark_amd as eb
  import torch
  from oracle.sawyer_oracle import SawyerPegOracle
  n, T, f = 5, 8, 3
  env = eb.EARLEnvs('sawyer_peg', reward_type='sparse', setup_as_lifelong_learning=True, reset_train_env_at_goal=True, num_envs=n, seed=4,
                    train_horizon=100, goal_change_frequency=f).get_envs()
  obs0 = env.reset().cpu().numpy()
  refs = [SawyerPegOracle(lm, 'sparse', True, seed=4, env_id=i, goal_change_frequency=f) for i in range(n)]
  for r in refs:
    r._settled = refs[0].settle()
    r.counter = 1
  np.testing.assert_allclose(obs0, np.stack([r.reset() for r in refs]), rtol=0, atol=1e-8)
  acts = np.random.default_rng(9).uniform(-1, 1, size=(T, n, 4)).astype(np.float32)
  out = env.rollout(torch.from_numpy(acts[:5]).cuda())
  rest = [env.step(torch.from_numpy(acts[t]).cuda()) for t in range(5, T)]                 # the counter carries across launches
  obs = np.concatenate([out['obs'].cpu().numpy()] + [o[0].cpu().numpy()[None] for o in rest])
  rew = np.concatenate([out['reward'].cpu().numpy()] + [o[1].cpu().numpy()[None] for o in rest])
  goals = [obs0[:, 7:]]
  for t in range(T):
    for i, r in enumerate(refs):
      o, rw, _, _ = r.step(acts[t, i])
      np.testing.assert_allclose(obs[t, i], o, rtol=0, atol=1e-7, err_msg=f'step {t} env {i}')
      assert float(rew[t, i]) == float(rw)
    goals.append(obs[t][:, 7:])
  changed = [bool((goals[t + 1] != goals[t]).any()) for t in range(T)]
  assert changed[2] and changed[5] and not any(changed[t] for t in (0, 1, 3, 4, 6, 7))       # switches on steps 3 and 6 only
  u = env.unwrapped
  assert u.steps_since_goal_change.tolist() == [2] * n
  np.testing.assert_allclose(u.goal_t.cpu().numpy(), goals[-1], atol=0)
  np.testing.assert_allclose(env.lifelong_return.cpu().numpy(), rew.sum(0), atol=1e-12)


def test_demo_prefixes_peg_drop_and_gripper():
  """first 12 steps of the 10 forward demonstrations from their recorded start (before the gripper reaches the peg): the peg is set
  down 5 mm above the table by reset_model and settles on it -- MuJoCo's recorded pegHead path is reproduced within 1.5 mm (free fall,
  soft table contact, friction), the gripper opening exactly (clipped at 1.0), the hand within 1.5 cm"""
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  eps = episodes('forward')
  n, K = len(eps), 12
  env = SawyerPeg(num_envs=n)
  env.reset()
  place_pegs(env, np.stack([e[0][4:7] for e in eps]).astype(np.float64))
  acts = np.stack([e[1][:K] for e in eps], axis=1)
  got = env.rollout(torch.from_numpy(acts).cuda())['obs'].cpu().numpy()
  want = np.stack([e[2][:K] for e in eps], axis=1)
  assert np.abs(got[..., 3] - want[..., 3]).max() < 1e-6
  assert np.abs(got[..., 4:7] - want[..., 4:7]).max() < 1.5e-3
  assert np.abs(got[..., :3] - want[..., :3]).max() < 1.5e-2
  assert abs(got[-1, :, 6].mean() - 0.015) < 1e-3 and abs(want[-1, :, 6].mean() - 0.015) < 1e-3       # both rest on the table top


def test_dense_reward_matches_the_restatement(lm):
  """reward_type='dense' (sawyer_peg.py:231-299): metaworld's tolerance(long_tail) / rect_prism_tolerance / hamacher_product /
  _gripper_caging_reward are UPSTREAM code that is not in the reference tree -- restated in oracle/sawyer_oracle.py, UNPINNED; this
  checks the kernel against that restatement through reaching, caging, the lifted branch (1 + 5 in_place) and success (10)"""
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from oracle.sawyer_oracle import SawyerPegOracle
  eps = episodes('forward')
  pick = [0, 3, 5, 9]                                   # episodes whose open-loop replay lifts and inserts the peg
  n = len(pick)
  env = SawyerPeg(reward_type='dense', num_envs=n, seed=6)
  env.reset()
  heads = np.stack([eps[i][0][4:7] for i in pick]).astype(np.float64)
  place_pegs(env, heads)
  env.obj_init[:, :3] = env.qpos[:, 9:12]; env.obj_init[:, 3:] = torch.from_numpy(heads).cuda()     # as reset_model would have left them
  refs = [SawyerPegOracle(lm, 'dense', seed=6, env_id=i) for i in range(n)]
  for i, r in enumerate(refs):
    r._settled = refs[0].settle(); r.counter = 1; r.reset()
    r.qpos[9:12] = heads[i] + [0.1, 0, 0]; r.obj_init_pos = r.qpos[9:12].copy(); r.peg_head_pos_init = heads[i].copy()
  np.testing.assert_allclose(np.ctypeslib.as_array(env._cfg.init_tcp), refs[0].init_tcp, atol=1e-9)
  T = max(len(eps[i][1]) for i in pick)
  acts = np.zeros((T, n, 4), np.float32)
  for j, i in enumerate(pick):
    acts[:len(eps[i][1]), j] = eps[i][1]
  seen, seen_info = set(), set()
  for t in range(T):
    o, r, done, info = env.step(torch.from_numpy(acts[t]).cuda())
    for j, ref in enumerate(refs):
      _, rr, _, ok = ref.step(acts[t, j])
      assert abs(float(r[j]) - float(rr)) <= 1e-6 * max(1.0, abs(float(rr))), (t, j, float(r[j]), float(rr))
      seen.add('ten' if rr == 10 else ('lifted' if rr > 1 else 'shaping'))
      # the info dict of SawyerPegV2.step (evaluate_state, sawyer_peg.py:165-184), every key
      for k, want in ref.last_info.items():
        assert abs(float(info[k][j]) - want) <= 1e-6 * max(1.0, abs(want)), (t, j, k, float(info[k][j]), want)
      seen_info.update(k for k in ('grasp_success', 'near_object', 'success') if ref.last_info[k] == 1.0)
      # resynchronise (contact dynamics are only piecewise smooth; the reward has thresholds)
    q = np.stack([ref.qpos for ref in refs]); v = np.stack([ref.qvel for ref in refs]); mp = np.stack([ref.mocap for ref in refs])
    env.qpos[:] = torch.from_numpy(q).cuda(); env.qvel[:] = torch.from_numpy(v).cuda(); env.mocap_pos[:] = torch.from_numpy(mp).cuda()
  assert seen == {'ten', 'lifted', 'shaping'}
  assert 'success' in seen_info             # (near_object / grasp_success compare the HAND with the grasp site, a finger length apart: they stay 0 in these replays)


def test_info_dict_with_the_sparse_reward_type(lm):
  """reward_type 'sparse': the reference still evaluates the shaping terms for its info dict, with object_grasped = 0 unless the peg is lifted
  (sawyer_peg.py:284-285) and unscaled_reward = float(is_successful) (:296-297)"""
  import torch
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from oracle.sawyer_oracle import SawyerPegOracle
  n, T = 3, 12
  env = SawyerPeg(reward_type='sparse', num_envs=n, seed=9)
  o0 = env.reset().cpu().numpy()
  refs = [SawyerPegOracle(lm, 'sparse', seed=9, env_id=i) for i in range(n)]
  for j, r_ in enumerate(refs):
    r_._settled = refs[0].settle(); r_.counter = 1                              # (the env's constructor has drawn the reset of counter 0)
    np.testing.assert_allclose(r_.reset(), o0[j], rtol=0, atol=1e-6)
  rng = np.random.default_rng(4)
  acts = rng.uniform(-1, 1, (T, n, 4)).astype(np.float32)
  acts[:, :, 2] = -np.abs(acts[:, :, 2])                                       # towards the table: the fingers reach the peg's height
  for t in range(T):
    o, r, done, info = env.step(torch.from_numpy(acts[t]).cuda())
    assert set(info) == set(_abi.SAWYER_INFO_KEYS) | {'is_successful', 'status'}
    for j, ref in enumerate(refs):
      ref.step(acts[t, j])
      for k, want in ref.last_info.items():
        assert abs(float(info[k][j]) - want) <= 1e-6 * max(1.0, abs(want)), (t, j, k, float(info[k][j]), want)
      assert float(info['grasp_reward'][j]) in (0.0, 1.0) and float(info['unscaled_reward'][j]) == float(r[j])
    q = np.stack([ref.qpos for ref in refs]); v = np.stack([ref.qvel for ref in refs]); mp = np.stack([ref.mocap for ref in refs])
    env.qpos[:] = torch.from_numpy(q).cuda(); env.qvel[:] = torch.from_numpy(v).cuda(); env.mocap_pos[:] = torch.from_numpy(mp).cuda()


def test_reset_records_the_state_the_dense_reward_needs():
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  env = SawyerPeg(reward_type='dense', num_envs=7, seed=2)
  o = env.reset()
  oi = env.obj_init.cpu().numpy()
  np.testing.assert_allclose(oi[:, :3], env.qpos[:, 9:12].cpu().numpy(), atol=0)       # obj_init_pos = the drawn peg position
  np.testing.assert_allclose(oi[:, 3:], o[:, 4:7].cpu().numpy(), atol=0)               # peg_head_pos_init = the pegHead site right after it


@pytest.mark.parametrize('task', ['sawyer_door', 'sawyer_peg', 'sawyer_peg:reset_at_goal', 'sawyer_peg:wide_init'])
def test_long_random_rollouts_stay_finite(task):
  """2,000 env steps (10,000 timesteps) of uniform random actions in 512 envs: no NaN / inf anywhere, the hand stays inside the mocap box
  (+ 10 cm: soft weld), the peg stays above the table and inside the walls' reach, quaternions stay normalised, velocities bounded"""
  import torch
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  n, T, chunks = 512, 250, 8
  task, _, mode = task.partition(':')
  env = (SawyerPeg if task == 'sawyer_peg' else SawyerDoor)(num_envs=n, seed=13, **({mode: True} if mode else {}))
  env.reset()
  g = torch.Generator(device='cuda').manual_seed(17)
  for c in range(chunks):
    acts = (torch.rand(T, n, 4, generator=g, device='cuda') * 2 - 1).float()
    out = env.rollout(acts)
    obs = out['obs']
    assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(env.qpos).all()) and bool(torch.isfinite(env.qvel).all()), c
    hand = obs[..., :3]
    lo = torch.tensor([-0.5, 0.40, 0.05], device='cuda') - 0.1
    hi = torch.tensor([0.5, 1.0, 0.5], device='cuda') + 0.1
    assert bool((hand >= lo).all()) and bool((hand <= hi).all()), c
    assert float(env.qvel.abs().max()) < 200.0, c
    if task == 'sawyer_peg':
      q = env.qpos
      assert float((q[:, 12:16].norm(dim=1) - 1).abs().max()) < 1e-9
      assert bool((q[:, 11] > -0.02).all()) and bool((q[:, 9:12].abs() < 2.0).all()), (c, float(q[:, 11].min()))
    else:
      assert bool((env.qpos[:, 9] > -2.2).all()) and bool((env.qpos[:, 9] < 0.3).all()), c        # door hinge range -1.57 .. 0 (+ soft limit)


def test_reverse_demos_open_loop_loose():
  """the 20 reverse demonstrations (reset_at_goal: the peg starts inside the hole, the policy pulls it out and lays it down at one of the
  initial states) replayed OPEN LOOP: the peg settling inside the hole during the first 12 steps follows MuJoCo's recording within 2.5 mm
  in at least 18 episodes (sphere chain / corner points against the hole walls), the hand path within 1.6 cm RMS in all, the peg path
  within 2 cm RMS over the WHOLE episode in all 20; the recorded episodes end ON the success radius (4.1 - 5.0 cm), so only some replays
  end inside it.  Measured this round with the calibrated weld: 19 / 20 prefixes within 1.7 mm, hand RMS 0.4-1.3 cm, peg RMS
  0.4-1.7 cm in 20 / 20 (derived weld: 14 / 20 under 2 cm, six lost the peg), 8 / 20 successes."""
  import torch
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  eps = episodes('reverse')
  n, T = len(eps), max(len(e[1]) for e in eps)
  assert n == 20
  env = SawyerPeg(num_envs=n, reset_at_goal=True)
  env.reset()
  place_pegs(env, np.stack([e[0][4:7] for e in eps]).astype(np.float64))
  env.goal_t[:] = torch.from_numpy(np.stack([e[0][7:] for e in eps]).astype(np.float64)).cuda()
  acts = np.zeros((T, n, 4), np.float32)
  for i, e in enumerate(eps):
    acts[:len(e[1]), i] = e[1]
  out = env.rollout(torch.from_numpy(acts).cuda())
  obs = out['obs'].cpu().numpy()
  assert np.isfinite(obs).all()
  prefix = path = 0
  for i, e in enumerate(eps):
    L = len(e[1])
    o, w = obs[:L, i], e[2]
    assert np.sqrt(((o[:, :3] - w[:, :3]) ** 2).sum(1).mean()) < 0.016, i
    prefix += np.abs(o[11, 4:7] - w[11, 4:7]).max() < 2.5e-3
    path += np.sqrt(((o[:, 4:7] - w[:, 4:7]) ** 2).sum(1).mean()) < 0.02
    np.testing.assert_allclose(o[:, 7:], np.repeat(e[0][7:][None].astype(np.float64), L, 0), atol=0)
  assert prefix >= 18 and path == 20, (prefix, path)
