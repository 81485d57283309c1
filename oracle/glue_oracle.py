"""numpy/ctypes front-end of oracle/glue_oracle.c (TEST INFRASTRUCTURE, NOT PRODUCT)."""
import ctypes as C

import numpy as np

from oracle.tabletop_oracle import lib, _p


class MotorParams(C.Structure):
  _fields_ = [('kp', C.c_double), ('kd', C.c_double), ('voltage', C.c_double), ('viscous_damping', C.c_double),
              ('torque_control', C.c_int32)]


def sawyer_sparse(obs, radius):
  obs = np.ascontiguousarray(obs)
  n = len(obs)
  r, s = np.zeros(n, np.float32), np.zeros(n, np.uint8)
  fn = lib().oracle_sawyer_sparse_f64 if obs.dtype == np.float64 else lib().oracle_sawyer_sparse_f32
  assert obs.dtype in (np.float64, np.float32) and obs.shape == (n, 14)
  fn(C.c_int32(n), _p(obs), C.c_double(radius), _p(r), _p(s))
  return r, s


def leg_to_motor(action):
  a = np.ascontiguousarray(action, np.float64)
  out = np.zeros_like(a)
  lib().oracle_minitaur_leg_to_motor(C.c_int32(len(a)), _p(a), _p(out))
  return out


def motor_torque(command, angle, velocity, kp=1.2, kd=0.0, voltage=16.0, viscous_damping=0.0, torque_control=False):
  c, a, v = (np.ascontiguousarray(x, np.float64) for x in (command, angle, velocity))
  act, obs = np.zeros_like(c), np.zeros_like(c)
  p = MotorParams(kp, kd, voltage, viscous_damping, int(torque_control))
  lib().oracle_minitaur_motor_torque(C.c_int32(c.size), C.byref(p), _p(c), _p(a), _p(v), _p(act), _p(obs))
  return act, obs


def minitaur_reward(obs, distance_weight=2.0, energy_weight=0.005, time_step=0.01):
  o = np.ascontiguousarray(obs, np.float64)
  r, s = np.zeros(len(o)), np.zeros(len(o), np.uint8)
  lib().oracle_minitaur_reward(C.c_int32(len(o)), _p(o), C.c_double(distance_weight), C.c_double(energy_weight),
                               C.c_double(time_step), _p(r), _p(s))
  return r, s


def kitchen_reward(obs, mocap_pos, site_xpos):
  o, mp, sx = (np.ascontiguousarray(x, np.float64) for x in (obs, mocap_pos, site_xpos))
  n = len(o)
  assert o.shape == (n, 46) and mp.shape == (n, 3) and sx.shape == (n, 8, 3)
  r, s = np.zeros(n), np.zeros(n, np.uint8)
  lib().oracle_kitchen_reward(C.c_int32(n), _p(o), _p(mp), _p(sx), _p(r), _p(s))
  return r, s.astype(bool)


class KitchenParams(C.Structure):   # struct earl_kitchen_params (include/earl_glue.h); values from the golden fixture (the reference's config)
  _fields_ = [('pos_bound', C.c_double * 2 * 23), ('vel_bound', C.c_double * 2 * 23), ('pos_noise_amp', C.c_double * 23),
              ('act_mid', C.c_double * 9), ('act_amp', C.c_double * 9), ('mocap_range', C.c_double * 3),
              ('mocap_clip_lower', C.c_double * 3), ('mocap_clip_upper', C.c_double * 3), ('step_duration', C.c_double),
              ('robot_noise_ratio', C.c_double)]


def kitchen_params(pos_bound, vel_bound, pos_noise_amp):
  """KitchenV0 constants (kitchen_multitask_v0.py:40-53, :78-79) + the joint table as the reference's Robot._read_specs_from_config returned it"""
  p = KitchenParams()
  for k in range(23):
    p.pos_bound[k][0], p.pos_bound[k][1] = float(pos_bound[k][0]), float(pos_bound[k][1])
    p.vel_bound[k][0], p.vel_bound[k][1] = float(vel_bound[k][0]), float(vel_bound[k][1])
    p.pos_noise_amp[k] = float(pos_noise_amp[k])
  for k in range(9):
    p.act_mid[k], p.act_amp[k] = 0.0, 2.0
  for k, (lo, hi) in enumerate(((-0.7, 0.4), (-0.1, 0.5), (1.8, 2.6))):
    p.mocap_range[k], p.mocap_clip_lower[k], p.mocap_clip_upper[k] = 0.01, lo, hi
  p.step_duration, p.robot_noise_ratio = 40 * 0.002, 0.1
  return p


def kitchen_action(p, action, mocap_pos, last_qpos_robot):
  a, lq = (np.ascontiguousarray(x, np.float64) for x in (action, last_qpos_robot))
  mp = np.array(mocap_pos, np.float64)
  ctrl = np.zeros_like(a)
  lib().oracle_kitchen_action(C.c_int32(len(a)), C.byref(p), _p(a), _p(mp), _p(lq), _p(ctrl))
  return mp, ctrl


def kitchen_obs(p, qpos, goal, noise=None):
  q, g = (np.ascontiguousarray(x, np.float64) for x in (qpos, goal))
  u = None if noise is None else np.ascontiguousarray(noise, np.float64)
  obs = np.zeros((len(q), 46))
  lib().oracle_kitchen_obs(C.c_int32(len(q)), C.byref(p), _p(q), _p(g), None if u is None else _p(u), _p(obs))
  return obs
