"""The `_cpu` entry points of include/earl_tabletop.h (BASELINE configs[0] "1 env, CPU ... plumbing, no GPU"): csrc/libearl_host.so is the
gfx950 kernels' OWN per-env functions (csrc/tabletop_device.h, tabletop_step.h, philox.h) compiled for the host by g++.  Checked here, without a GPU,
against (a) every tabletop golden recorded from the reference's classes and all 2,534 demonstration transitions, bit for bit, (b) the C oracle on seeded
batches through every mode (reset modes, lifelong switching, auto-reset, 3-object variant, NaN / inf / out-of-range actions), (c) the loader:
`EARLEnvs(..., num_envs=1, device='cpu')` returns the reference's scalar gym 4-tuple (envs/tabletop_manipulation.py:128-138).  The host library is a
product build (no oracle code in it: tests/test_abi.py); the oracle stays the checker."""
import os
import re
import subprocess

import numpy as np
import pytest

import hip_harness as hx
from conftest import REPO, load_golden
from oracle import tabletop_oracle as orc
from test_oracle import DENSE_ATOL, DENSE_RTOL, np122

CPU = 'cpu'


def host(n, **kw):
  return hx.HipTabletop(n, device=CPU, **kw)


def _inject(h, g, n):
  import torch
  h.goal_table = torch.tensor(np.ascontiguousarray(g['goal'], np.float64))
  h.cfg.n_goals = n
  h.goal_idx.copy_(torch.arange(n, dtype=torch.int32))
  h.qpos.copy_(torch.from_numpy(g['qpos0']))
  h.attached.copy_(torch.from_numpy(g['attached0'].astype(np.int8)))


def test_library_exports_every_cpu_symbol_and_needs_no_hip_runtime():
  from earl_benchmark_amd import _abi
  lib = _abi.load_host()
  assert lib.earl_version().startswith(b'earl-host')
  for name in _abi.HOST_SIGNATURES:
    assert hasattr(lib, name)
  assert {'earl_tabletop_step_cpu', 'earl_tabletop_reset_cpu', 'earl_tabletop_rollout_cpu'} <= set(_abi.HOST_SIGNATURES)
  # declared in the header, one `_cpu` twin per device entry point of the tabletop path
  hdr = open(os.path.join(REPO, 'include', 'earl_tabletop.h')).read()
  declared = set(re.findall(r'\bint (earl_tabletop3?_\w+_cpu)\(', hdr))
  assert declared == set(_abi.HOST_SIGNATURES), declared ^ set(_abi.HOST_SIGNATURES)
  needed = subprocess.run(['ldd', _abi.HOST_LIB_PATH], capture_output=True, text=True).stdout
  assert 'amdhip' not in needed and 'hsa' not in needed and 'oracle' not in needed


def test_onestep_goldens_bit_exact():
  g = load_golden('tabletop_onestep')
  n = len(g['qpos0'])
  for rt, key in (('sparse', 'reward_sparse'), ('dense', 'reward_dense')):
    h = host(n, reward_type=rt, horizon=10**9)
    _inject(h, g, n)
    obs, rew, done, succ = h.step(g['action'])
    np.testing.assert_array_equal(h.host('qpos'), g['qpos1'])
    np.testing.assert_array_equal(h.host('attached'), g['attached1'])
    np.testing.assert_array_equal(obs.view(np.uint32), g['obs'].view(np.uint32))
    np.testing.assert_array_equal(succ.astype(bool), g['success'])
    assert not done.any()
    if rt == 'sparse':
      np.testing.assert_array_equal(rew, g[key])
    else:
      np.testing.assert_allclose(rew, g[key], rtol=DENSE_RTOL, atol=DENSE_ATOL)
  h = host(n, reward_type='sparse', wide_init=True, horizon=10**9)
  _inject(h, g, n)
  obs, rew, done, succ = h.step(g['action'])
  np.testing.assert_array_equal(succ.astype(bool), g['success_wide'])
  np.testing.assert_array_equal(rew, g['reward_sparse_wide'])
  r, s = hx.hip_reward(g['obs'], 'sparse', device=CPU)
  np.testing.assert_array_equal(r, g['reward_sparse'])
  np.testing.assert_array_equal(s.astype(bool), g['success'])
  r, s = hx.hip_reward(g['obs'], 'dense', device=CPU)
  np.testing.assert_allclose(r, g['reward_dense'], rtol=DENSE_RTOL, atol=DENSE_ATOL)


@pytest.mark.parametrize('rt', ['sparse', 'dense'])
def test_rollout_goldens(rt):
  g = load_golden('tabletop_rollouts')
  acts = g['actions']
  R, T, _ = acts.shape
  h = host(R, reward_type=rt, horizon=int(g['horizon']))
  obs0 = h.reset(next_goal_idx=g['goal_idx'])
  np.testing.assert_array_equal(obs0, g[f'{rt}_obs0'])
  obs, rew, done, succ = h.rollout(np.ascontiguousarray(acts.transpose(1, 0, 2)))
  np.testing.assert_array_equal(obs.transpose(1, 0, 2), g[f'{rt}_obs'])
  np.testing.assert_array_equal(done.T.astype(bool), g[f'{rt}_done'])
  want = np122(g[f'{rt}_success'], g[f'{rt}_norm4'])
  np.testing.assert_array_equal(succ.T.astype(bool), want)
  np.testing.assert_array_equal(h.host('qpos'), g[f'{rt}_qpos'][:, -1])
  if rt == 'sparse':
    np.testing.assert_array_equal(rew.T, want.astype(np.float32))
  else:
    np.testing.assert_allclose(rew.T, g[f'{rt}_reward'], rtol=DENSE_RTOL, atol=DENSE_ATOL)
  assert (h.host('num_interventions') == 1).all() and (h.host('steps_since_reset') == T).all()


def test_wide_init_and_lifelong_goldens():
  g = load_golden('tabletop_wide_init')
  np.testing.assert_array_equal(hx.hip_valid_init(g['candidates'], device=CPU).astype(bool), g['valid'])
  g = load_golden('tabletop_lifelong')
  for rt in ('sparse', 'dense'):
    T = len(g[f'{rt}_actions'])
    h = host(1, reward_type=rt, horizon=int(g['train_horizon']), goal_change_frequency=int(g['freq']))
    obs0 = h.reset(next_goal_idx=[int(g[f'{rt}_goal0'])])
    np.testing.assert_array_equal(obs0[0], g[f'{rt}_obs0'])
    for t in range(T):
      ob, rw, dn, sc = h.step(g[f'{rt}_actions'][t][None], next_goal_idx=[int(g[f'{rt}_goal_seq'][t])])
      np.testing.assert_array_equal(ob[0], g[f'{rt}_obs'][t])
      assert bool(dn[0]) == bool(g[f'{rt}_done'][t])
      if rt == 'sparse':
        assert rw[0] == g[f'{rt}_reward'][t] and h.host('lifelong_return')[0] == g[f'{rt}_return'][t]
      else:
        np.testing.assert_allclose(rw[0], g[f'{rt}_reward'][t], rtol=DENSE_RTOL, atol=DENSE_ATOL)


@pytest.mark.parametrize('direction', ['forward', 'reverse'])
def test_demonstrations_replay(direction):
  """all 2,534 transitions the reference ships (recorded upstream with the MuJoCo-backed class)"""
  import torch
  demo = np.load(os.path.join(REPO, 'earl_benchmark_amd', 'demonstrations', 'tabletop_manipulation', direction, 'demo_data.npz'))
  rep = load_golden('tabletop_demo_replay')
  ob, act = demo['observations'], demo['actions']
  n = len(ob)
  h = host(n, horizon=10**9)
  h.goal_table = torch.tensor(ob[:, 6:12].astype(np.float64))
  h.cfg.n_goals = n
  h.goal_idx.copy_(torch.arange(n, dtype=torch.int32))
  h.qpos.copy_(torch.from_numpy(ob[:, :4].astype(np.float64)))
  h.attached.copy_(torch.from_numpy(ob[:, 4].astype(np.int8)))
  obs, rew, done, succ = h.step(act)
  assert np.abs(obs - demo['next_observations']).max() < 5e-7
  np.testing.assert_array_equal(rew, demo['rewards'][:, 0])
  np.testing.assert_array_equal(obs, rep[f'{direction}_next_obs'])
  np.testing.assert_array_equal(rew, rep[f'{direction}_reward'])
  np.testing.assert_array_equal(h.host('attached'), rep[f'{direction}_attached'])


def test_3obj_goldens():
  import torch
  g = load_golden('tabletop3_onestep')
  n = len(g['qpos0'])
  for rt in ('sparse', 'dense'):
    h = host(n, reward_type=rt, horizon=10**9, nobj=3, goal_table=g['goal'])
    h.goal_idx.copy_(torch.arange(n, dtype=torch.int32))
    h.qpos.copy_(torch.from_numpy(g['qpos0'])); h.attached.copy_(torch.from_numpy(g['attached0'].astype(np.int8)))
    obs, rew, done, succ = h.step(g['action'])
    np.testing.assert_array_equal(h.host('qpos'), g['qpos1'])
    np.testing.assert_array_equal(h.host('attached'), g['attached1'])
    np.testing.assert_array_equal(obs, g['obs'])
    np.testing.assert_array_equal(succ.astype(bool), g['success'])
    if rt == 'sparse':
      np.testing.assert_array_equal(rew, g['reward_sparse'])
    else:
      np.testing.assert_allclose(rew, g['reward_dense'], rtol=DENSE_RTOL, atol=DENSE_ATOL)
    acts = g['roll_actions']
    R, T, _ = acts.shape
    h = host(R, reward_type=rt, horizon=10**9, nobj=3)
    np.testing.assert_array_equal(h.reset(), g[f'roll_{rt}_obs0'])
    obs, rew, done, succ = h.rollout(np.ascontiguousarray(acts.transpose(1, 0, 2)))
    np.testing.assert_array_equal(obs.transpose(1, 0, 2), g[f'roll_{rt}_obs'])


def _same(h, o):
  for k in h.STATE:
    np.testing.assert_array_equal(h.host(k), getattr(o, k), err_msg=k)
  assert h.cfg.counter == o.cfg.counter


@pytest.mark.parametrize('mode', ['fixed', 'at_goal', 'wide'])
@pytest.mark.parametrize('rt', ['sparse', 'dense'])
@pytest.mark.parametrize('n', [1, 63, 1500])
def test_seeded_batches_vs_oracle(mode, rt, n):
  """reset + fused rollout + single steps + masked reset, special actions included: every output and every state array equal to the oracle's, bit for
  bit (dense reward too: both sides are host code calling the same libm)"""
  T = 37
  kw = dict(reward_type=rt, reset_at_goal=mode == 'at_goal', wide_init=mode == 'wide', seed=11, env_offset=3, horizon=T - 5)
  rng = np.random.default_rng(n)
  o, h = orc.OracleTabletop(n, **kw), host(n, **kw)
  np.testing.assert_array_equal(h.reset(), o.reset())
  acts = rng.uniform(-1.3, 1.3, size=(T, n, 3)).astype(np.float32)
  acts[..., 2] = np.where(rng.random((T, n)) < 0.7, np.abs(acts[..., 2]), acts[..., 2])
  acts[T // 2, 0, 0] = np.nan
  acts[0, n // 2, 1] = np.inf
  acts[3, n - 1, 2] = -0.0
  for got, want in zip(h.rollout(acts), o.rollout(acts)):
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
  _same(h, o)
  m = (rng.random(n) < 0.5).astype(np.uint8)
  np.testing.assert_array_equal(h.reset(mask=m).view(np.uint32), o.reset(mask=m).view(np.uint32))
  for got, want in zip(h.step(acts[1]), o.step(acts[1])):
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
  for got, want in zip(h.rollout(acts[:5], reset_first=True), (o.reset(), o.rollout(acts[:5]))[1]):
    np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
  _same(h, o)
  for got, want in zip(h.observe(), o.observe()):
    if want is not None:
      np.testing.assert_array_equal(got.view(np.uint8), np.asarray(want).astype(got.dtype).view(np.uint8))


def test_lifelong_and_auto_reset_vs_oracle():
  n, T = 257, 120
  rng = np.random.default_rng(5)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  for kw in (dict(goal_change_frequency=7, horizon=50), dict(auto_reset=True, horizon=13), dict(auto_reset=True, wide_init=True, horizon=9, reward_type='dense'),
             dict(goal_change_frequency=5, auto_reset=True, horizon=11)):
    o, h = orc.OracleTabletop(n, seed=2, **kw), host(n, seed=2, **kw)
    np.testing.assert_array_equal(h.reset(), o.reset())
    for got, want in zip(h.rollout(acts), o.rollout(acts)):
      np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
    for t in range(3):
      for got, want in zip(h.step(acts[t]), o.step(acts[t])):
        np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
    _same(h, o)


def test_3obj_vs_oracle():
  n, T = 300, 40
  rng = np.random.default_rng(8)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  acts[..., 2] = np.abs(acts[..., 2])
  for kw in (dict(), dict(reset_at_goal=True, reward_type='dense')):
    o, h = orc.OracleTabletop(n, nobj=3, seed=4, horizon=30, **kw), host(n, nobj=3, seed=4, horizon=30, **kw)
    np.testing.assert_array_equal(h.reset(), o.reset())
    for got, want in zip(h.rollout(acts), o.rollout(acts)):
      np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
    _same(h, o)


def test_bad_arguments_are_errors_not_crashes():
  import ctypes as C
  from earl_benchmark_amd import _abi
  lib = _abi.load_host()
  h = host(4)
  st = h._state()
  assert lib.earl_tabletop_step_cpu(C.byref(h.cfg), C.byref(st), None, None, None) == -1
  assert b'NULL' in lib.earl_last_error()
  h.cfg.reward_type = 9
  arrs, out = h._outs((4,))
  assert lib.earl_tabletop_step_cpu(C.byref(h.cfg), C.byref(st), h.qpos.data_ptr(), None, C.byref(out)) == -1
  h.cfg.reward_type, h.cfg.n = 0, 0
  assert lib.earl_tabletop_step_cpu(C.byref(h.cfg), C.byref(st), h.qpos.data_ptr(), None, C.byref(out)) == 0     # n = 0: validated, nothing done
  assert np.isnan(arrs[0].numpy()).all()


def test_loader_scalar_env_on_the_host_is_the_reference_4_tuple():
  """BASELINE configs[0]: tabletop sparse, 1 env, CPU: `EARLEnvs(...).get_envs()` -> the wrapped scalar env; step() returns (ndarray[12] float32, python
  float, python bool, {}) like envs/tabletop_manipulation.py:128-138 under PersistentStateWrapper; one scripted evaluation episode equals the golden
  rollout recorded from the reference's own classes"""
  import earl_benchmark_amd as eb
  g = load_golden('tabletop_rollouts')
  loader = eb.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=1, device='cpu')
  train_env, eval_env = loader.get_envs()
  assert eval_env.unwrapped.scalar_api and str(eval_env.unwrapped.device) == 'cpu'
  want = np122(g['sparse_success'], g['sparse_norm4'])
  for r in range(4):
    obs = eval_env.reset(goal_idx=[int(g['goal_idx'][r])])
    assert isinstance(obs, np.ndarray) and obs.shape == (12,) and obs.dtype == np.float32
    np.testing.assert_array_equal(obs, g['sparse_obs0'][r])
    for t in range(g['actions'].shape[1]):
      obs, rew, done, info = eval_env.step(g['actions'][r, t])
      assert isinstance(obs, np.ndarray) and type(rew) is float and type(done) is bool and info == {}
      np.testing.assert_array_equal(obs, g['sparse_obs'][r, t])
      assert rew == float(want[r, t]) and done == bool(g['sparse_done'][r, t])
      assert eval_env.is_successful() == bool(want[r, t])
  assert eval_env.num_interventions == 4 and train_env.num_interventions == 0
  # batched on the host as well (same code path as device='cuda', other library)
  _, ev = eb.EARLEnvs('tabletop_manipulation', reward_type='dense', num_envs=64, device='cpu', seed=3).get_envs()
  o = ev.reset()
  assert tuple(o.shape) == (64, 12) and o.device.type == 'cpu'
  ob, rw, dn, info = ev.step(np.zeros((64, 3), np.float32))
  assert tuple(rw.shape) == (64,) and 'success' in info


def test_default_device_is_still_the_gpu_and_still_fails_loudly_without_one():
  import torch
  if torch.cuda.is_available():
    pytest.skip('a GPU is present')
  from earl_benchmark_amd import _abi
  from earl_benchmark_amd.envs import tabletop
  with pytest.raises(_abi.EarlHipError):
    tabletop.TabletopManipulation(num_envs=4)                  # no silent fallback to the host build
  with pytest.raises(_abi.EarlHipError):
    tabletop.TabletopManipulation(num_envs=4, device='cuda')
