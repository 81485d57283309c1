"""Batched, GPU-resident counterpart of the reference's minitaur env (BASELINE configs[4]).

Mirrors `GoalConditionedMinitaurBulletEnv` (reference: earl_benchmark/envs/minitaur_gym_env.py:466-546 on MinitaurBulletEnv :56-464 and
envs/minitaur.py: Minitaur): `reset`, `step`, `reset_goal`, `get_next_goal`, `compute_reward`, `is_successful`, `_get_obs`, the observation layout
(motor angles 8, velocities 8, torques 8, base orientation 4 (x, y, z, w), base xy 2, goal 2), dense reward, `done` never set by the env -- for
`num_envs` independent instances, one env step = ONE kernel launch (earl_minitaur_rollout with T = 1; `rollout` fuses T steps).

The numpy the reference wraps around Bullet is followed line by line in the kernel (leg model, velocity-limited commands, DC-motor model with
overheat protection, reward, success; the first three are pinned bit-exact by tests/test_glue*.py against goldens recorded from the reference's own
functions).  **THE RIGID-BODY PART IS UNPINNED AND MODEL-LESS**: the reference simulates pybullet_data's minitaur.urdf in PyBullet 3.2.0, neither of
which is in its tree; the robot model here (tools/minitaur_model.py: 22 dofs, four loop closures, spheres against the ground and the wall tiles) is
this build's own authoring on this build's own stepper (MuJoCo-style soft constraints, not Bullet's sequential-impulse solver).  The env randomizer
[UPSTREAM pybullet_envs.bullet.minitaur_env_randomizer, restated from memory] is built in the reset kernel: battery voltage, motor damping, base /
leg-link / motor masses, foot friction, through the semantics of the reference's own setters (envs/minitaur.py:468-508).  DESIGN.md section 14.
"""
import ctypes as C

import numpy as np
import torch

from .. import _abi, physics
from ..spaces import Box

INT32_MAX = 2**31 - 1
NUM_SUBSTEPS, SETTLE_STEPS = 5, 100                       # minitaur_gym_env.py:25, 161-164; :265-269
MOTOR_KP, MOTOR_KD, MOTOR_VELOCITY_LIMIT = 1.0, 0.02, 150.0   # :85-86, :472
DISTANCE_WEIGHT, ENERGY_WEIGHT = 2.0, 0.005               # :473, :71
OVERHEAT_SHUTDOWN_TORQUE, OVERHEAT_SHUTDOWN_TIME = 2.45, 1.0   # minitaur.py:14-15
ACTION_BOUND, ACTION_EPS = 1.0, 0.01                      # minitaur_gym_env.py:144, 30
GOAL_LOCATIONS = np.array([[0.4, 0.2], [0.2, 0.2], [-0.2, 0.2], [-0.4, 0.2], [0.4, 0.0], [0.2, 0.0], [-0.2, 0.0], [-0.4, 0.0],
                           [0.4, 0.4], [0.2, 0.4], [-0.2, 0.4], [-0.4, 0.4]])   # :467-469
OBS_DIM, ACT_DIM = 32, 8
BASE_MASS_ERR, LEG_MASS_ERR, FOOT_FRICTION = (-0.2, 0.2), (-0.2, 0.2), (0.8, 1.5)   # [UPSTREAM] the randomizer's ranges
N_PARAM = 6                                              # earl_minitaur_state.motor_param row: voltage, damping, three mass factors, foot friction


def make_cfg(tables, n=0, env_offset=0, horizon=0, randomize=True, seed=0, goal_table_ptr=None, reset_qpos_ptr=None):
  """struct earl_minitaur_cfg from the model tables (shared with the oracle's front end, which passes host pointers)"""
  dt = float(tables['timestep'])
  cfg = _abi.MinitaurCfg(n=n, env_offset=env_offset, horizon=horizon, num_substeps=NUM_SUBSTEPS, settle_steps=SETTLE_STEPS, randomize=(7 if randomize is True else (int(randomize) if randomize else 0)),
                         n_goals=len(GOAL_LOCATIONS), goal_change_frequency=0, overheat_steps=int(OVERHEAT_SHUTDOWN_TIME / dt),
                         motor_kp=MOTOR_KP, motor_kd=MOTOR_KD, motor_velocity_limit=MOTOR_VELOCITY_LIMIT, overheat_torque=OVERHEAT_SHUTDOWN_TORQUE,
                         distance_weight=DISTANCE_WEIGHT, energy_weight=ENERGY_WEIGHT, success_radius=0.1, goal_table=goal_table_ptr,
                         reset_qpos=reset_qpos_ptr, seed=int(seed) & (2**64 - 1), counter=0, step_counter=0)
  cfg.motor_dof[:] = [int(x) for x in tables['motor_dof']]
  cfg.base_mass_err[:], cfg.leg_mass_err[:], cfg.foot_friction[:] = BASE_MASS_ERR, LEG_MASS_ERR, FOOT_FRICTION
  cfg.leg_mass, cfg.motor_mass = float(tables['rand_leg_mass']), float(tables['rand_motor_mass'])
  cfg.motor_dir[:] = [float(x) for x in tables['motor_direction']]
  return cfg


class _Cfg(_abi.MinitaurCfg):
  pass


class Minitaur:
  OBS_DIM, NV, NQ = OBS_DIM, 22, 23

  def __init__(self, num_envs=1, device='cuda', seed=0, env_offset=0, scalar_api=None, env_randomizer=True, contacts=True, reset_at_goal=False,
               auto_reset=False, reward_type='dense'):
    if auto_reset or reset_at_goal:
      raise NotImplementedError('minitaur: auto_reset / reset_at_goal do not exist in the reference env')
    del reward_type                                          # (the reference's env has one reward: the loader's reward_type is not passed to it)
    self._lib = _abi.load()
    dev = torch.device(device)
    if dev.type != 'cuda' or not torch.cuda.is_available():
      raise _abi.EarlHipError(f'device={device!r}: the minitaur env runs on MI355X only (no CPU fallback)')
    if dev.index is None:
      dev = torch.device('cuda', torch.cuda.current_device())
    self.device, self.num_envs = dev, int(num_envs)
    n = self.num_envs
    self.scalar_api = (n == 1) if scalar_api is None else bool(scalar_api)
    if self.scalar_api and n != 1:
      raise ValueError('scalar_api needs num_envs == 1')
    with torch.cuda.device(dev):
      self.model = physics.DeviceModel('minitaur', device=dev, contacts=contacts)
    assert self.model.nv == self.NV and self.model.nq == self.NQ
    kw = dict(dtype=torch.float64, device=dev)
    self.qpos, self.qvel = torch.zeros(n, self.NQ, **kw), torch.zeros(n, self.NV, **kw)
    self.goal_t = torch.tensor(GOAL_LOCATIONS[0], **kw).repeat(n, 1).contiguous()
    self.motor_param = torch.tensor([16.0, 0.0, 1.0, 1.0, 1.0, -1.0], **kw).repeat(n, 1).contiguous()   # [n, N_PARAM], written by the reset kernel
    self.observed_torque = torch.zeros(n, 8, **kw)
    self.overheat = torch.zeros(n, 8, dtype=torch.int32, device=dev)
    self.motor_enabled = torch.ones(n, 8, dtype=torch.uint8, device=dev)
    self.steps_since_reset = torch.zeros(n, dtype=torch.int32, device=dev)
    self.steps_since_goal_change = torch.zeros(n, dtype=torch.int32, device=dev)
    self.interventions = torch.zeros(n, dtype=torch.int32, device=dev)
    self.fail_count = torch.zeros(n, dtype=torch.int32, device=dev)
    self.lifelong_return_t = torch.zeros(n, **kw)
    self.last_obs = torch.zeros(n, self.OBS_DIM, **kw)
    self.total_step_count = 0
    self._goal_table = torch.tensor(GOAL_LOCATIONS, **kw).contiguous()
    self._reset_qpos = torch.tensor(self.model.tables['qpos0'], **kw).contiguous()
    base = make_cfg(self.model.tables, n=n, env_offset=int(env_offset), horizon=INT32_MAX, randomize=env_randomizer, seed=seed,
                    goal_table_ptr=self._goal_table.data_ptr(), reset_qpos_ptr=self._reset_qpos.data_ptr())
    self._cfg = _Cfg.from_buffer_copy(base)
    self._st = _abi.MinitaurState(qpos=self.qpos.data_ptr(), qvel=self.qvel.data_ptr(), goal=self.goal_t.data_ptr(), motor_param=self.motor_param.data_ptr(),
                                  observed_torque=self.observed_torque.data_ptr(), overheat=self.overheat.data_ptr(), motor_enabled=self.motor_enabled.data_ptr(),
                                  steps_since_reset=self.steps_since_reset.data_ptr(), steps_since_goal_change=self.steps_since_goal_change.data_ptr(),
                                  fail_count=self.fail_count.data_ptr(), last_obs=self.last_obs.data_ptr())
    self.action_space = Box(-ACTION_BOUND, ACTION_BOUND, (ACT_DIM,), np.float32)     # minitaur_gym_env.py:175-178
    self.observation_space = Box(-np.inf, np.inf, (self.OBS_DIM,), np.float32)        # :179, :481-488
    self._counter = 0
    self._last_success = torch.zeros(n, dtype=torch.bool, device=dev)
    self.reset()
    self.interventions.zero_()

  # ------------------------------------------------------------------ internals
  @property
  def unwrapped(self):
    return self

  def _stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  def _new_out(self, lead):
    kw = dict(device=self.device)
    return dict(obs=torch.empty(*lead, self.num_envs, self.OBS_DIM, dtype=torch.float64, **kw), reward=torch.empty(*lead, self.num_envs, dtype=torch.float64, **kw),
                done=torch.empty(*lead, self.num_envs, dtype=torch.bool, **kw), success=torch.empty(*lead, self.num_envs, dtype=torch.bool, **kw),
                status=torch.empty(*lead, self.num_envs, dtype=torch.uint8, **kw))

  # ------------------------------------------------------------------ gym-style API
  def reset(self, mask=None):
    """GoalConditionedMinitaurBulletEnv.reset (:476-479) of the (masked) envs -> obs [N, 32] (numpy [32] with scalar_api)"""
    n = self.num_envs
    with torch.cuda.device(self.device):
      m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
      obs = self.last_obs.clone()
      self._cfg.counter = self._counter
      _abi.check(self._lib.earl_minitaur_reset(self.model.buf.data_ptr(), self.model.col_ptr, C.byref(self._cfg), C.byref(self._st),
                                               None if m is None else m.data_ptr(), obs.data_ptr(), self._stream()), 'earl_minitaur_reset')
      self.interventions += 1 if m is None else m.to(torch.int32)
    self._counter += 1
    return obs[0].cpu().numpy() if self.scalar_api else obs

  def _actions(self, a, lead):
    a = torch.as_tensor(np.asarray(a, dtype=np.float32) if not torch.is_tensor(a) else a, device=self.device).to(torch.float32)
    a = a.reshape(*lead, self.num_envs, ACT_DIM).contiguous()
    # minitaur_gym_env.py:276-281: `if not (-bound - eps <= a_i <= bound + eps): raise` -- the NEGATED in-range test, so a NaN raises too (the
    # mirrored out-of-range test let NaN through; the motor model then carried it into the state and the env sat in its failure guard)
    out_of_bounds = ~((a >= -ACTION_BOUND - ACTION_EPS) & (a <= ACTION_BOUND + ACTION_EPS))
    if bool(out_of_bounds.any()):
      bad = int(torch.nonzero(out_of_bounds.reshape(-1, ACT_DIM).any(0))[0])
      raise ValueError('{}th action out of bounds.'.format(bad))
    return a

  def rollout(self, actions, out=None):
    """T env steps in ONE launch: actions [T, N, 8] -> dict(obs [T,N,32], reward [T,N], done, success, status); bit-identical to T step() calls"""
    with torch.cuda.device(self.device):
      T = int(torch.as_tensor(actions).shape[0])
      a = self._actions(actions, (T,))
      res = out if out is not None else self._new_out((T,))
      o = _abi.MinitaurOut(obs=res['obs'].data_ptr(), reward=res['reward'].data_ptr(), done=res['done'].data_ptr(), success=res['success'].data_ptr(),
                           status=res['status'].data_ptr())
      self._cfg.step_counter = self.total_step_count
      if T > 0:
        _abi.check(self._lib.earl_minitaur_rollout(self.model.buf.data_ptr(), self.model.col_ptr, C.byref(self._cfg), C.byref(self._st), a.data_ptr(), T,
                                                   C.byref(o), self._stream()), 'earl_minitaur_rollout')
        if int(self._cfg.goal_change_frequency) > 0:
          self.lifelong_return_t += res['reward'].sum(0)
        self._last_success = res['success'][-1]
    self.total_step_count += T
    return res

  def step(self, action):
    """-> (obs [N,32], reward [N], done [N], info{success, status}); gym 4-tuple of numpy / python scalars with scalar_api"""
    res = self.rollout(torch.as_tensor(np.asarray(action, dtype=np.float32) if not torch.is_tensor(action) else action).reshape(1, self.num_envs, ACT_DIM))
    obs, rew, done, suc = res['obs'][0], res['reward'][0], res['done'][0], res['success'][0]
    if self.scalar_api:
      return obs[0].cpu().numpy(), float(rew[0]), bool(done[0]), {'success': float(suc[0])}
    return obs, rew, done, {'success': suc, 'status': res['status'][0]}

  def _get_obs(self):
    """GetObservation + goal of the CURRENT state (:541-546): no simulation, the newest observed torques"""
    md, dr = torch.tensor([int(x) for x in self.model.tables['motor_dof']], device=self.device), torch.tensor(self.model.tables['motor_direction'], device=self.device)
    ang = self.qpos[:, md + 1] * dr
    vel = self.qvel[:, md] * dr
    q = self.qpos[:, 3:7]
    obs = torch.cat([ang, vel, self.observed_torque, q[:, 1:4], q[:, 0:1], self.qpos[:, 0:2], self.goal_t], 1)
    return obs[0].cpu().numpy() if self.scalar_api else obs

  get_obs = _get_obs

  def compute_reward(self, obs):
    """GoalConditionedMinitaurBulletEnv.compute_reward (:529-535)"""
    o = torch.as_tensor(obs, dtype=torch.float64, device=self.device).reshape(-1, self.OBS_DIM)
    dist = -(o[:, 28] - o[:, 30]).abs() - (o[:, 29] - o[:, 31]).abs()
    energy = (o[:, 8:16] * o[:, 16:24]).sum(1).abs() * float(self.model.tables['timestep'])
    r = DISTANCE_WEIGHT * dist - ENERGY_WEIGHT * energy
    return float(r[0]) if self.scalar_api else r

  def is_successful(self, obs=None):
    o = self._get_obs() if obs is None else obs
    o = torch.as_tensor(o, dtype=torch.float64, device=self.device).reshape(-1, self.OBS_DIM)
    s = ((o[:, 28:30] - o[:, 30:32]) ** 2).sum(1).sqrt() < 0.1                    # :495-503
    return float(s[0]) if self.scalar_api else s

  # ------------------------------------------------------------------ goals (:481-493)
  def get_next_goal(self):
    return GOAL_LOCATIONS[np.random.randint(len(GOAL_LOCATIONS))].copy()

  def reset_goal(self, goal=None, mask=None):
    g = self.get_next_goal() if goal is None else np.asarray(goal, np.float64)
    g = g[-2:] if g.shape[-1] == 30 else g                                          # :484-487
    g = torch.as_tensor(g, dtype=torch.float64, device=self.device).expand(self.num_envs, 2)
    if mask is None:
      self.goal_t.copy_(g)
    else:
      m = torch.as_tensor(mask, device=self.device).bool()
      self.goal_t[m] = g[m]

  @property
  def goal(self):
    return self.goal_t[0].cpu().numpy() if self.scalar_api else self.goal_t

  def set_state(self, qpos, qvel):
    self.qpos.copy_(torch.as_tensor(qpos, dtype=torch.float64, device=self.device).reshape(self.num_envs, self.NQ))
    self.qvel.copy_(torch.as_tensor(qvel, dtype=torch.float64, device=self.device).reshape(self.num_envs, self.NV))

  _STATE = ('qpos', 'qvel', 'goal_t', 'motor_param', 'observed_torque', 'overheat', 'motor_enabled', 'steps_since_reset', 'steps_since_goal_change',
            'interventions', 'fail_count', 'lifelong_return_t', 'last_obs')

  def state_dict(self):
    return {k: getattr(self, k).clone() for k in self._STATE} | {'counter': self._counter, 'total_step_count': self.total_step_count}

  def load_state_dict(self, sd):
    for k, v in sd.items():
      if k == 'counter':
        self._counter = int(v)
      elif k == 'total_step_count':
        self.total_step_count = int(v)
      else:
        getattr(self, k).copy_(v)
