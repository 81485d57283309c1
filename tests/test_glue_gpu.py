"""HIP glue kernels (include/earl_glue.h) vs the oracle and the goldens: bit-exact (fp64 arithmetic, no libm calls
other than correctly-rounded sqrt)."""
import numpy as np
import pytest

import earl_benchmark_amd as eb
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def t():
  import torch
  return torch


def dev(t, a):
  return t.from_numpy(np.ascontiguousarray(a)).cuda()


def test_sawyer_sparse(t):
  from earl_benchmark_amd import glue
  from oracle import glue_oracle as go
  g = load_golden('physics_glue')
  for name in ('sawyer_door', 'sawyer_peg'):
    rew, suc = glue.sawyer_sparse_reward(dev(t, g['sawyer_obs']), name)
    np.testing.assert_array_equal(suc.cpu().numpy(), g[f"{name}_success"])
    np.testing.assert_array_equal(rew.cpu().numpy(), g[f"{name}_success"].astype(np.float32))
    L = eb.EARLEnvs.__new__(eb.EARLEnvs); L._env_name = name
    for demo in L.get_demonstrations():          # all 2,910 recorded rows, float32 as shipped
      rew, suc = glue.sawyer_sparse_reward(dev(t, demo['next_observations']), name)
      np.testing.assert_array_equal(rew.cpu().numpy(), demo['rewards'][:, 0])
  # ulp walk around the radius: device sqrt must be correctly rounded to agree with the oracle everywhere
  obs = np.zeros((4000, 14)); x = 0.02
  for _ in range(2000):
    x = np.nextafter(x, 0)
  for i in range(4000):
    obs[i, 4] = x; x = np.nextafter(x, 1)
  rew, suc = glue.sawyer_sparse_reward(dev(t, obs), 'sawyer_door')
  np.testing.assert_array_equal(suc.cpu().numpy().astype(np.uint8), go.sawyer_sparse(obs, 0.02)[1])
  assert 0 < int(suc.sum()) < 4000


def test_minitaur_leg_motor_reward(t):
  from earl_benchmark_amd import glue
  from oracle import glue_oracle as go
  g = load_golden('physics_glue')
  np.testing.assert_array_equal(glue.leg_to_motor(dev(t, g['leg_actions'])).cpu().numpy(), g['leg_motor_angles'])
  for tag in ('pos', 'pd', 'torque'):
    kp, kd, tc, volt, visc = g[f'motor_{tag}_params']
    cmd = g['motor_torque_cmd'] if tag == 'torque' else g['motor_cmd']
    act, obs = glue.motor_torque(dev(t, cmd), dev(t, g['motor_angle']), dev(t, g['motor_vel']), kp, kd, volt, visc, bool(tc))
    np.testing.assert_array_equal(act.cpu().numpy(), g[f'motor_{tag}_actual'])
    np.testing.assert_array_equal(obs.cpu().numpy(), g[f'motor_{tag}_observed'])
  rew, suc = glue.minitaur_reward(dev(t, g['minitaur_obs']))
  np.testing.assert_array_equal(rew.cpu().numpy(), g['minitaur_reward'])
  np.testing.assert_array_equal(suc.cpu().numpy(), g['minitaur_success'].astype(bool))
  # bigger random batch vs the oracle (interp table boundaries included)
  rng = np.random.default_rng(0)
  m = 200001
  cmd, ang = rng.uniform(-3, 3, m), rng.uniform(-3, 3, m)
  vel = rng.uniform(-700, 700, m)
  vel[:7] = (16.0 * np.clip(-1.2 * (ang[:7] - cmd[:7]), -1, 1) - 0.186 * np.arange(0, 70, 10)) / 0.0954   # current on the table knots
  act, obs = glue.motor_torque(dev(t, cmd), dev(t, ang), dev(t, vel))
  a0, o0 = go.motor_torque(cmd, ang, vel)
  np.testing.assert_array_equal(act.cpu().numpy(), a0); np.testing.assert_array_equal(obs.cpu().numpy(), o0)


def test_glue_rejects_cpu_tensors(t):
  from earl_benchmark_amd import _abi, glue
  with pytest.raises(_abi.EarlHipError):
    glue.leg_to_motor(t.zeros(2, 8, dtype=t.float64))


def test_kitchen_reward_hip_matches_the_reference_goldens(t):
  from earl_benchmark_amd import glue
  z = load_golden('kitchen_glue')
  r, s = glue.kitchen_reward(dev(t, z['kitchen_obs']), dev(t, z['kitchen_mocap']), dev(t, z['kitchen_site_xpos']))
  assert (r.cpu().numpy() == z['kitchen_reward']).all() and (s.cpu().numpy() == z['kitchen_success']).all()
  assert list(glue.KITCHEN_SITES) == [str(x) for x in z['kitchen_site_names']]


def test_kitchen_step_and_obs_glue_hip_match_the_reference_goldens(t):
  """earl_kitchen_action / earl_kitchen_obs (rows a17, a18) against values recorded from the reference's own methods; the default
  parameter table compiled into the library equals what the reference reads from its franka_config.xml"""
  from earl_benchmark_amd import glue
  z = load_golden('kitchen_step')
  p = glue.kitchen_params()
  assert (np.ctypeslib.as_array(p.pos_bound) == z['kitchen_pos_bound']).all() and (np.ctypeslib.as_array(p.vel_bound) == z['kitchen_vel_bound']).all()
  assert (np.ctypeslib.as_array(p.pos_noise_amp) == z['kitchen_pos_noise_amp']).all() and p.step_duration == 40 * 0.002
  mp = dev(t, z['kstep_mocap'].copy())
  ctrl = glue.kitchen_action(dev(t, z['kstep_action']), mp, dev(t, z['kstep_last_qpos']))
  assert (mp.cpu().numpy() == z['kstep_new_mocap']).all() and (ctrl.cpu().numpy() == z['kstep_ctrl']).all()
  mp = dev(t, z['kstep_mocap'].copy())
  ctrl = glue.kitchen_action(dev(t, z['kstep_action'].astype(np.float32)), mp, dev(t, z['kstep_last_qpos']))
  assert (mp.cpu().numpy() == z['kstep_new_mocap_f32act']).all() and (ctrl.cpu().numpy() == z['kstep_ctrl_f32act']).all()
  obs = glue.kitchen_obs(dev(t, z['kobs_qpos']), dev(t, z['kobs_goal']), dev(t, z['kobs_uniform']))
  assert (obs.cpu().numpy() == z['kobs_obs']).all()
  clean = glue.kitchen_obs(dev(t, z['kobs_qpos']), dev(t, z['kobs_goal']))
  assert (clean.cpu().numpy() == np.concatenate([z['kobs_qpos'], z['kobs_goal']], 1)).all()
