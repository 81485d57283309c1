"""Batched (GPU) versions of the pure-numpy glue of EARL's physics-backed envs -- include/earl_glue.h.

The dynamics of sawyer_door / sawyer_peg / kitchen / minitaur are MuJoCo / Bullet in the reference and this build's own stepper here (include/earl_physics.h,
parity unpinned: DESIGN.md sections 9-11, 14); these are the functions the reference itself evaluates in numpy around the dynamics, at parity with the reference
(tests/test_glue.py, tests/test_glue_gpu.py):
  sawyer_sparse_reward  SawyerDoorV2 / SawyerPegV2 .is_successful and the sparse compute_reward branch
  leg_to_motor          Minitaur.ConvertFromLegModel
  motor_torque          MotorModel.convert_to_torque
  minitaur_reward       GoalConditionedMinitaurBulletEnv.compute_reward / is_successful
All take and return torch tensors on a HIP device.
"""
import ctypes as C

import torch

from . import _abi

SAWYER_RADIUS = {'sawyer_door': 0.02, 'sawyer_peg': 0.05}   # sawyer_door.py:177, sawyer_peg.py:63 (TARGET_RADIUS)


def _stream(t):
  return torch.cuda.current_stream(t.device).cuda_stream


def _cuda(t, dtype=None):
  t = torch.as_tensor(t)
  if not t.is_cuda:
    raise _abi.EarlHipError('glue functions run on MI355X only: pass tensors on a HIP device')
  if dtype is not None and t.dtype != dtype:
    t = t.to(dtype)
  return t.contiguous()


def sawyer_sparse_reward(obs, env_name='sawyer_door', radius=None):
  """obs [n, 14] float64 (env) or float32 (demonstration layout) -> (reward [n] float32, success [n] bool)."""
  lib = _abi.load()
  obs = _cuda(obs)
  if obs.dtype not in (torch.float32, torch.float64) or obs.ndim != 2 or obs.shape[1] != 14:
    raise ValueError('obs must be [n, 14] float32/float64')
  r = SAWYER_RADIUS[env_name] if radius is None else float(radius)
  n = obs.shape[0]
  rew = torch.empty(n, dtype=torch.float32, device=obs.device)
  suc = torch.empty(n, dtype=torch.bool, device=obs.device)
  fn = lib.earl_sawyer_sparse_f64 if obs.dtype == torch.float64 else lib.earl_sawyer_sparse_f32
  with torch.cuda.device(obs.device):
    _abi.check(fn(n, obs.data_ptr(), r, rew.data_ptr(), suc.data_ptr(), _stream(obs)), 'sawyer_sparse')
  return rew, suc


def leg_to_motor(action):
  """action [n, 8] -> desired motor angles [n, 8] float64."""
  lib = _abi.load()
  a = _cuda(action, torch.float64).reshape(-1, 8)
  out = torch.empty_like(a)
  with torch.cuda.device(a.device):
    _abi.check(lib.earl_minitaur_leg_to_motor(a.shape[0], a.data_ptr(), out.data_ptr(), _stream(a)), 'leg_to_motor')
  return out


def motor_torque(command, angle, velocity, kp=1.2, kd=0.0, voltage=16.0, viscous_damping=0.0, torque_control=False):
  """-> (actual_torque, observed_torque), same shape as `command` (float64)."""
  lib = _abi.load()
  c, a, v = _cuda(command, torch.float64), _cuda(angle, torch.float64), _cuda(velocity, torch.float64)
  if not (c.shape == a.shape == v.shape):
    raise ValueError('command, angle, velocity must have the same shape')
  act, obs = torch.empty_like(c), torch.empty_like(c)
  p = _abi.MotorParams(kp, kd, voltage, viscous_damping, int(bool(torque_control)))
  with torch.cuda.device(c.device):
    _abi.check(lib.earl_minitaur_motor_torque(c.numel(), C.byref(p), c.data_ptr(), a.data_ptr(), v.data_ptr(),
                                              act.data_ptr(), obs.data_ptr(), _stream(c)), 'motor_torque')
  return act, obs


def minitaur_reward(obs, distance_weight=2.0, energy_weight=0.005, time_step=0.01):
  """obs [n, 32] -> (reward [n] float64, success [n] bool)."""
  lib = _abi.load()
  o = _cuda(obs, torch.float64).reshape(-1, 32)
  n = o.shape[0]
  rew = torch.empty(n, dtype=torch.float64, device=o.device)
  suc = torch.empty(n, dtype=torch.bool, device=o.device)
  with torch.cuda.device(o.device):
    _abi.check(lib.earl_minitaur_reward(n, o.data_ptr(), distance_weight, energy_weight, time_step, rew.data_ptr(),
                                        suc.data_ptr(), _stream(o)), 'minitaur_reward')
  return rew, suc


KITCHEN_SITES = ('knob1_site', 'knob2_site', 'knob3_site', 'knob4_site', 'light_site', 'slide_site', 'hinge_site2', 'microhandle_site')


def kitchen_reward(obs, mocap_pos, site_xpos):
  """Kitchen.compute_reward / is_successful (kitchen.py:141-183): obs [n,46], mocap_pos [n,3], site_xpos [n,8,3] in
  KITCHEN_SITES order -> (reward [n] float64, success [n] bool)."""
  lib = _abi.load()
  o = _cuda(obs, torch.float64).reshape(-1, 46)
  n = o.shape[0]
  mp = _cuda(mocap_pos, torch.float64).reshape(n, 3)
  sx = _cuda(site_xpos, torch.float64).reshape(n, 8, 3)
  rew = torch.empty(n, dtype=torch.float64, device=o.device)
  suc = torch.empty(n, dtype=torch.bool, device=o.device)
  with torch.cuda.device(o.device):
    _abi.check(lib.earl_kitchen_reward(n, o.data_ptr(), mp.data_ptr(), sx.data_ptr(), rew.data_ptr(), suc.data_ptr(), _stream(o)),
               'kitchen_reward')
  return rew, suc


def kitchen_params():
  """struct earl_kitchen_params with the reference's values (franka_config.xml joint table, KitchenV0 step constants)"""
  p = _abi.KitchenParams()
  _abi.check(_abi.load().earl_kitchen_default_params(C.byref(p)), 'kitchen_default_params')
  return p


def kitchen_action(action, mocap_pos, last_qpos_robot, params=None):
  """KitchenV0.step up to do_simulation (kitchen_multitask_v0.py:91-105, franka_robot.py:172-207, :259-264): action [n,9], mocap_pos
  [n,3] (updated IN PLACE), last_qpos_robot [n,9] -> ctrl [n,9] (the position targets handed to do_simulation)"""
  lib = _abi.load()
  p = params or kitchen_params()
  a = _cuda(action, torch.float64).contiguous()
  n = a.shape[0]
  lq = _cuda(last_qpos_robot, torch.float64).contiguous()
  assert a.shape == (n, 9) and lq.shape == (n, 9) and mocap_pos.shape == (n, 3) and mocap_pos.dtype == torch.float64 and mocap_pos.is_contiguous()
  ctrl = torch.empty(n, 9, dtype=torch.float64, device=a.device)
  with torch.cuda.device(a.device):
    _abi.check(lib.earl_kitchen_action(n, C.byref(p), a.data_ptr(), mocap_pos.data_ptr(), lq.data_ptr(), ctrl.data_ptr(), _stream(a)), 'kitchen_action')
  return ctrl


def kitchen_obs(qpos, goal, noise=None, params=None):
  """Robot.get_obs + KitchenV0._get_obs (franka_robot.py:137-168, kitchen_multitask_v0.py:127-139): qpos [n,23], goal [n,23], noise [n,46]
  (the U(-1, 1) draws of one get_obs in call order) or None -> obs [n,46]"""
  lib = _abi.load()
  p = params or kitchen_params()
  q, g = _cuda(qpos, torch.float64).contiguous(), _cuda(goal, torch.float64).contiguous()
  n = q.shape[0]
  u = None if noise is None else _cuda(noise, torch.float64).contiguous()
  assert q.shape == (n, 23) and g.shape == (n, 23) and (u is None or u.shape == (n, 46))
  obs = torch.empty(n, 46, dtype=torch.float64, device=q.device)
  with torch.cuda.device(q.device):
    _abi.check(lib.earl_kitchen_obs(n, C.byref(p), q.data_ptr(), g.data_ptr(), None if u is None else u.data_ptr(), obs.data_ptr(), _stream(q)),
               'kitchen_obs')
  return obs
