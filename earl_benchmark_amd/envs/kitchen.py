"""Batched, GPU-resident counterpart of the reference's kitchen env (BASELINE configs[3]).

Mirrors `Kitchen(KitchenTaskRelaxV1)` (reference: earl_benchmark/envs/kitchen.py:86-187 on top of
envs/kitchen_assets/adept_envs/adept_envs/franka/kitchen_multitask_v0.py:91-139 and .../franka/robot/franka_robot.py:137-264): `reset`, `step`,
`reset_goal`, `get_next_goal`, `compute_reward`, `is_successful`, `_get_obs`, observation layout (robot qpos 9, fixture qpos 14, goal 23; with
the reference's uniform sensor noise), dense reward only -- for `num_envs` independent instances.

One env step = ONE C-ABI call, earl_kitchen_step (include/earl_physics.h), which puts eight launches on the caller's stream, among them:
  earl_kitchen_action  (csrc/glue.hip)    action clip / scale, mocap target, the nine position targets (row a17; bit-exact vs the reference's numpy)
  the nv = 23 stepper  (csrc/physics.hip) 40 timesteps: 32 lanes per env, tree-structured in-LDS factorisations                        (row a16)
  earl_kitchen_obs     (csrc/glue.hip)    observation with sensor noise from Philox draws                                              (row a18)
  earl_kitchen_reward  (csrc/glue.hip)    Kitchen._get_reward_n_score / is_successful                                                  (row a19)
plus the per-env failure guard and the wrapper bookkeeping as small kernels.

STATUS of the dynamics: this build's own articulated-body stepper on tables compiled from the reference's MJCF (tools/mjcf_compile.py kitchen):
Franka arm (link inertias from the collision hulls and the given masses), 14 single-dof fixtures, the mocap weld on panda0_link7, joint limits,
the five knob / burner and switch / light couplings, dry friction and springs on the fixture joints, force-limited finger actuators, and a
DECLARED, reduced collision set (each finger one box; handles as sphere chains; door panels against the finger corners; since round 3 hand, wrist,
forearm (13 spheres) and the finger corners against the counter top, oven body, back wall, hood, microwave body and cabinet bottoms; links 1-4, the
right counter and the floor are not collided).  **PARITY WITH MUJOCO IS UNPINNED**: the reference ships no recording of this env and MuJoCo cannot
run here.  What is pinned: the numpy glue around the simulator (bit-exact on goldens recorded from the reference's own methods),
the model tables' provenance, and the kernel against this build's CPU statement (oracle/physics_oracle.LinkModel).
"""
import ctypes as C

import numpy as np
import torch

from .. import _abi, glue, physics, tables
from ..spaces import Box

INT32_MAX = 2**31 - 1
FRAME_SKIP = 40                                           # kitchen_multitask_v0.py:40
MIDPOINT_POS = (-0.440, 0.1, 2.226)                       # :46
# kitchen_multitask_v0.py:68-73 (overrides the keyframe)
INIT_QPOS = np.array([1.48388023e-01, -1.76848573e+00, 1.84390296e+00, -2.47685760e+00, 2.60252026e-01, 7.12533105e-01, 1.59515394e+00,
                      4.79267505e-02, 3.71350919e-02, -2.66279850e-04, -5.18043486e-05, 3.12877220e-05, -4.51199853e-05, -3.90842156e-06,
                      -4.22629655e-05, 6.28065475e-05, 4.04984708e-05, 4.62730939e-04, -2.26906415e-04, -4.65501369e-04, -6.44129196e-03,
                      -1.77048263e-03, 1.08009684e-03])
STREAM_NOISE, STREAM_RESET = 0x4B00, 0x4B80               # Philox stream ids (csrc/glue.hip earl_philox_uniform)
# Robot.get_obs's velocity noise amplitudes (reference adept_envs/franka/robot/franka_config.xml:17-43, attribute vel_noise_amp of qpos0 ... qpos22: the
# robot's nine joints 0.1; the fixtures 0.005, the three hinge joints at the end 0.1); the position amplitudes are in earl_kitchen_params
VEL_NOISE_AMP = np.array([0.1] * 9 + [0.005] * 11 + [0.1] * 3)
RESET_ROBOT_STEPS = 10                                    # KitchenV0.reset_model: ten zero-action robot steps after sim.reset() (kitchen_multitask_v0.py:150-153)


class _Cfg(_abi.KitchenCfg):
  """struct earl_kitchen_cfg + the lifelong wrapper's number (wrappers.py sets `horizon` and `goal_change_frequency` the way it does for the other envs)"""
  goal_change_frequency = 0


class Kitchen:
  OBS_DIM, NV, N_ROBOT, N_OBJ = 46, 23, 9, 14

  def __init__(self, task='all_pairs', reward_type='dense', num_envs=1, device='cuda', seed=0, env_offset=0, scalar_api=None,
               sensor_noise=True, contacts=True, reset_at_goal=False, auto_reset=False, info='full'):
    """info: 'full' (default) = step() returns the reference's env_info (time, obs_dict incl. the noisy velocity readings, rewards, score, images);
    'minimal' = only 'success' / 'is_successful' / 'status' (batched) or {} (scalar): no extra Philox draw, concatenation or host copies per step"""
    if info not in ('full', 'minimal'):
      raise ValueError(f"info must be 'full' or 'minimal', got {info!r}")
    self.info_mode = info
    if reward_type != 'dense':
      raise ValueError('Kitchen environment only supports dense rewards.')      # kitchen.py:91-92
    if auto_reset or reset_at_goal:
      raise NotImplementedError('kitchen: auto_reset / reset_at_goal do not exist in the reference env')
    self._lib = _abi.load()
    dev = torch.device(device)
    if dev.type != 'cuda' or not torch.cuda.is_available():
      raise _abi.EarlHipError(f'device={device!r}: the kitchen env runs on MI355X only (no CPU fallback)')
    if dev.index is None:
      dev = torch.device('cuda', torch.cuda.current_device())
    self.device, self.num_envs, self._task = dev, int(num_envs), task
    n = self.num_envs
    self.scalar_api = (n == 1) if scalar_api is None else bool(scalar_api)
    if self.scalar_api and n != 1:
      raise ValueError('scalar_api needs num_envs == 1')
    self._seed, self._env_offset, self._counter = int(seed) & (2**64 - 1), int(env_offset), 0
    self._fused_step = True            # step() through earl_kitchen_rollout(T = 1); False: through earl_kitchen_step (the per-step C entry point, kept and tested)
    self.sensor_noise = bool(sensor_noise)
    self._params = glue.kitchen_params()
    self._initial_states = tables.initial_states('kitchen')                     # kitchen.py:57-85, the 6 'all_pairs' rows (get_init_states, :103-104)
    try:                                                                          # reset_model reads initial_states[task] (:122-126): the four single-fixture
      self._task_rows = np.atleast_2d(tables.get(f'kitchen_task_{task}'))       # tasks, the six pairs and 'all_pairs' (tables.npz, from the reference module)
    except KeyError:
      raise KeyError(f'kitchen task {task!r}: the reference defines ' + ', '.join(tables.kitchen_tasks())) from None
    self._goal_states = tables.goal_states('kitchen')
    with torch.cuda.device(dev):
      self.model = physics.DeviceModel('kitchen', device=dev, contacts=contacts)
    assert self.model.nv == self.NV
    names = self.model.att_names
    self._site_idx = torch.tensor([names.index(s) for s in glue.KITCHEN_SITES], device=dev)
    kw = dict(dtype=torch.float64, device=dev)
    self.qpos, self.qvel = torch.zeros(n, self.NV, **kw), torch.zeros(n, self.NV, **kw)
    self.mocap_pos = torch.tensor(MIDPOINT_POS, **kw).repeat(n, 1).contiguous()
    self.mocap_quat = torch.tensor(self.model.tables['weld_mocap_quat'], **kw).repeat(n, 1).contiguous()   # the mocap body's orientation never changes
    self.goal_t = torch.tensor(self._goal_states[0], **kw).repeat(n, 1).contiguous()
    self.last_qp_robot = torch.zeros(n, self.N_ROBOT, **kw)                     # qpos_robot of the newest cached (noisy) observation
    self.att = torch.zeros(n, self.model.n_att, 3, **kw)
    self.steps_since_reset = torch.zeros(n, dtype=torch.int32, device=dev)
    self.interventions = torch.zeros(n, dtype=torch.int32, device=dev)
    self.fail_count = torch.zeros(n, dtype=torch.int32, device=dev)
    self.lifelong_return_t = torch.zeros(n, **kw)
    self.steps_since_goal_change = torch.zeros(n, dtype=torch.int32, device=dev)
    self.total_step_count = 0
    self.last_obs = torch.zeros(n, self.OBS_DIM, **kw)
    # scratch of earl_kitchen_step (caller-owned like the state) and its argument structs
    self._scr = dict(action64=torch.zeros(n, 9, **kw), ctrl9=torch.zeros(n, 9, **kw), noise=torch.zeros(n, 46, **kw), qpos_bak=torch.zeros(n, self.NV, **kw),
                     qvel_bak=torch.zeros(n, self.NV, **kw), sites=torch.zeros(n, 8, 3, **kw), bad=torch.zeros(n, dtype=torch.uint8, device=dev),
                     mocap_bak=torch.zeros(n, 3, **kw), att_bak=torch.zeros(n, self.model.n_att, 3, **kw))
    self._mq1 = self.mocap_quat[0].clone().contiguous()
    self._cfg = _Cfg(n=n, env_offset=self._env_offset, horizon=INT32_MAX, frame_skip=FRAME_SKIP, sensor_noise=int(self.sensor_noise), n_att=self.model.n_att,
                     seed=self._seed, counter=0, mocap_quat_dev=self._mq1.data_ptr())
    self._cfg.site_att[:] = [names.index(s) for s in glue.KITCHEN_SITES]
    self._st = _abi.KitchenState(qpos=self.qpos.data_ptr(), qvel=self.qvel.data_ptr(), mocap_pos=self.mocap_pos.data_ptr(), goal=self.goal_t.data_ptr(),
                                 last_qp_robot=self.last_qp_robot.data_ptr(), att_xpos=self.att.data_ptr(), steps_since_reset=self.steps_since_reset.data_ptr(),
                                 fail_count=self.fail_count.data_ptr(), last_obs=self.last_obs.data_ptr(), **{k: v.data_ptr() for k, v in self._scr.items()})
    self.action_space = Box(-1.0, 1.0, (self.N_ROBOT,), np.float32)              # kitchen_multitask_v0.py:78-80
    self.observation_space = Box(-8.0, 8.0, (self.OBS_DIM,), np.float64)         # :82-84
    with torch.cuda.device(dev):
      self._reset_states = self._settle_reset_states()
      self.reset()
    self.interventions.zero_()

  # ------------------------------------------------------------------ internals
  @property
  def unwrapped(self):
    return self

  def _stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  def _uniform(self, k, stream_id, lo, hi, n=None):
    n = self.num_envs if n is None else n
    out = torch.empty(n, k, dtype=torch.float64, device=self.device)
    with torch.cuda.device(self.device):
      _abi.check(self._lib.earl_philox_uniform(n, k, self._seed, self._counter, self._env_offset, stream_id, lo, hi, out.data_ptr(), self._stream()),
                 'earl_philox_uniform')
    return out

  def _settle_reset_states(self):
    """Kitchen.reset_model (kitchen.py:118-139) for each of the initial-state rows: robot.reset (qpos <- init_qpos with the row's fixture
    positions, clipped to the position bounds; qvel <- 0), mocap <- midpoint, then ten robot steps of zero action = 400 timesteps.  Those
    steps are deterministic: their position targets are the cached robot joints 0 and 1 clamped to the finger actuators' ctrlrange [0, 0.04]
    (the nu = 2 quirk), i.e. 0.04 and 0 whatever the sensor noise -- so the result is computed once per row and cached."""
    rows = self._task_rows                      # kitchen.py:122-126: a draw over the six 'all_pairs' rows, or the ONE row of a named task
    kw = dict(dtype=torch.float64, device=self.device)
    pb = np.ctypeslib.as_array(self._params.pos_bound)
    q = np.tile(INIT_QPOS, (len(rows), 1))
    q[:, 9:] = rows[:, 9:]
    q = np.clip(q, pb[:, 0], pb[:, 1])                                           # Robot.reset -> clip_positions (franka_robot.py:212, :170-174)
    q, v = torch.tensor(q, **kw).contiguous(), torch.zeros(len(rows), self.NV, **kw)
    mp = torch.tensor(MIDPOINT_POS, **kw).repeat(len(rows), 1).contiguous()
    mq = self.mocap_quat[:1].repeat(len(rows), 1).contiguous()
    ctrl = torch.tensor([[0.04, 0.0]], **kw).repeat(len(rows), 1).contiguous()
    self.model.step(q, v, mp, mq, ctrl, nsub=10 * FRAME_SKIP)
    return q, v

  def _observe(self, noise):
    u = self._uniform(46, STREAM_NOISE, -1.0, 1.0) if (noise and self.sensor_noise) else None
    obs = glue.kitchen_obs(self.qpos, self.goal_t, u, self._params)
    self.last_qp_robot.copy_(obs[:, :self.N_ROBOT])
    return obs

  def _reward(self, obs):
    sites = self.att.index_select(1, self._site_idx).contiguous()
    return glue.kitchen_reward(obs, self.mocap_pos, sites)

  # ------------------------------------------------------------------ gym-style API
  def reset(self, mask=None):
    """reset (masked) envs -> obs [N, 46] (numpy [46] with scalar_api)"""
    n = self.num_envs
    with torch.cuda.device(self.device):
      rows = self._reset_states[0].shape[0]
      pick = (self._uniform(1, STREAM_RESET, 0.0, 1.0)[:, 0] * rows).long().clamp_(max=rows - 1)      # np.random.randint(rows), kitchen.py:123
      m = torch.ones(n, dtype=torch.bool, device=self.device) if mask is None else torch.as_tensor(mask, device=self.device).bool()
      self.qpos[m] = self._reset_states[0][pick][m]
      self.qvel[m] = self._reset_states[1][pick][m]
      self.mocap_pos[m] = torch.tensor(MIDPOINT_POS, dtype=torch.float64, device=self.device)
      self.goal_t[m] = torch.tensor(self.get_next_goal(), dtype=torch.float64, device=self.device)   # reset_goal(), kitchen.py:138
      self.steps_since_reset[m] = 0
      self.steps_since_goal_change[m] = 0
      self.interventions += m.to(torch.int32)
      # set_state -> sim.forward(): site positions of the reset state (nothing integrated)
      self.att.copy_(self.model.forward(self.qpos, self.qvel, self.mocap_pos, self.mocap_quat, torch.zeros(n, 2, dtype=torch.float64, device=self.device))[2])
      prev = self.last_obs.clone()
      obs = self._observe(noise=True)
      obs = torch.where(m[:, None], obs, prev)
      self.last_obs.copy_(obs)
    self._counter += 1
    return obs[0].cpu().numpy() if self.scalar_api else obs

  def step(self, action, b=None, out=None):
    """one env step of every env: ONE launch (the fused rollout kernel with T = 1; `_fused_step = False` takes earl_kitchen_step's eight launches instead)"""
    del b
    n = self.num_envs
    with torch.cuda.device(self.device):
      a = torch.as_tensor(np.asarray(action, dtype=np.float32) if not torch.is_tensor(action) else action, device=self.device)
      a = a.to(torch.float32).reshape(n, self.N_ROBOT).contiguous()
      if out is None:
        out = dict(obs=torch.empty(n, self.OBS_DIM, dtype=torch.float64, device=self.device), reward=torch.empty(n, dtype=torch.float64, device=self.device),
                   done=torch.empty(n, dtype=torch.bool, device=self.device), success=torch.empty(n, dtype=torch.bool, device=self.device),
                   status=torch.empty(n, dtype=torch.uint8, device=self.device))
      o = _abi.KitchenOut(obs=out['obs'].data_ptr(), reward=out['reward'].data_ptr(), done=out['done'].data_ptr(), success=out['success'].data_ptr(),
                          status=out['status'].data_ptr())
      self._cfg.counter = self._counter
      if self._fused_step:             # ONE launch: the fused rollout kernel with T = 1 (bit-identical to earl_kitchen_step's eight launches, tests/test_kitchen_gpu.py)
        _abi.check(self._lib.earl_kitchen_rollout(self.model.buf.data_ptr(), self.model.col_ptr, C.byref(self._params), C.byref(self._cfg), C.byref(self._st),
                                                  a.data_ptr(), 1, C.byref(o), self._stream()), 'earl_kitchen_rollout')
      else:
        _abi.check(self._lib.earl_kitchen_step(self.model.buf.data_ptr(), self.model.col_ptr, C.byref(self._params), C.byref(self._cfg), C.byref(self._st),
                                               a.data_ptr(), C.byref(o), self._stream()), 'earl_kitchen_step')
      obs, rew, done, suc = out['obs'], out['reward'], out['done'], out['success']
      gcf = int(self._cfg.goal_change_frequency)
      if gcf > 0:                                                                # LifelongWrapper.step (lifelong_wrapper.py:30-44)
        self.lifelong_return_t += rew
        self.steps_since_goal_change += 1
        sw = self.steps_since_goal_change >= gcf
        self.steps_since_goal_change[sw] = 0
        self.goal_t[sw] = torch.tensor(self.get_next_goal(), dtype=torch.float64, device=self.device)
        obs = torch.cat([obs[:, :23], torch.where(sw[:, None], self.goal_t, obs[:, 23:])], 1)
    if self.info_mode == 'full':
      info = self._env_info(obs, rew, suc, out['status'], self._counter)
    else:
      info = {} if self.scalar_api else {'success': suc, 'is_successful': suc, 'status': out['status']}
    self._counter += 1
    self.total_step_count += 1
    self._last_success = suc
    if self.scalar_api:
      return obs[0].cpu().numpy(), float(rew[0]), bool(done[0]), info
    return obs, rew, done, info

  def _env_info(self, obs, rew, suc, status, counter):
    """The env_info dict of KitchenV0.step (reference adept_envs/franka/kitchen_multitask_v0.py:116-123): 'time' (simulation time of the observation: sim.reset()
    at the last reset, ten robot steps there, then one per step), 'obs_dict' (t, qp, qv, obj_qp, obj_qv, goal: Robot.get_obs's noisy readings, franka_robot.py:137-168 --
    the velocities get draws 9-17 and 32-45 of the step's 46 uniforms, the positions are the observation's), 'rewards' (Kitchen._get_reward_n_score, envs/kitchen.py:141-175:
    true_reward = r_total = the step's reward), 'score' (0.), 'images' ([]); plus this build's own 'success' / 'is_successful' (is_successful of every env) and 'status'."""
    dt = float(self.model.tables['timestep'])
    t = (RESET_ROBOT_STEPS + self.steps_since_reset.to(torch.float64)) * (FRAME_SKIP * dt)
    qv = self.qvel
    if self.sensor_noise:
      c = self._counter
      self._counter = counter                              # the draws of THIS step (earl_kitchen_cfg.counter): the same Philox stream the observation used
      try:
        u = self._uniform(46, STREAM_NOISE, -1.0, 1.0)
      finally:
        self._counter = c
      amp = torch.as_tensor(float(self._params.robot_noise_ratio) * VEL_NOISE_AMP, dtype=torch.float64, device=self.device)
      qv = qv + amp * torch.cat([u[:, 9:18], u[:, 32:46]], 1)
    if self.scalar_api:
      h = torch.cat([obs[0], qv[0], t[:1], rew[:1].to(torch.float64)]).cpu().numpy()      # one copy to the host: obs 46, qv 23, t, reward
      od = {'t': float(h[69]), 'qp': h[:9].copy(), 'qv': h[46:55].copy(), 'obj_qp': h[9:23].copy(), 'obj_qv': h[55:69].copy(), 'goal': h[23:46].copy()}
      r = float(h[70])
      return {'time': od['t'], 'obs_dict': od, 'rewards': {'true_reward': r, 'r_total': r}, 'score': 0.0, 'images': []}
    od = {'t': t, 'qp': obs[:, :9], 'qv': qv[:, :9], 'obj_qp': obs[:, 9:23], 'obj_qv': qv[:, 9:], 'goal': obs[:, 23:]}
    return {'time': t, 'obs_dict': od, 'rewards': {'true_reward': rew, 'r_total': rew}, 'score': torch.zeros_like(rew), 'images': [],
            'success': suc, 'is_successful': suc, 'status': status}

  def rollout(self, actions, out=None):
    """T steps: actions [T, N, 9] -> dict(obs [T,N,46], reward [T,N], done, success, status).  ONE launch (earl_kitchen_rollout: every wave walks its
    envs through all T steps) unless the lifelong wrapper's goal switch is on, which steps; the results are those of T step() calls, bit for bit."""
    a = torch.as_tensor(actions, device=self.device)
    T = a.shape[0]
    if int(self._cfg.goal_change_frequency) == 0 and not self.scalar_api and T > 0:
      n = self.num_envs
      with torch.cuda.device(self.device):
        a = a.to(torch.float32).reshape(T, n, self.N_ROBOT).contiguous()
        res = out if out is not None else {}
        kw = dict(device=self.device)
        for k, shape, dt in (('obs', (T, n, self.OBS_DIM), torch.float64), ('reward', (T, n), torch.float64), ('done', (T, n), torch.bool),
                             ('success', (T, n), torch.bool), ('status', (T, n), torch.uint8)):
          if k not in res or res[k].shape != shape or res[k].dtype != dt:
            res[k] = torch.empty(shape, dtype=dt, **kw)
        o = _abi.KitchenOut(obs=res['obs'].data_ptr(), reward=res['reward'].data_ptr(), done=res['done'].data_ptr(), success=res['success'].data_ptr(),
                            status=res['status'].data_ptr())
        self._cfg.counter = self._counter
        _abi.check(self._lib.earl_kitchen_rollout(self.model.buf.data_ptr(), self.model.col_ptr, C.byref(self._params), C.byref(self._cfg), C.byref(self._st),
                                                  a.data_ptr(), T, C.byref(o), self._stream()), 'earl_kitchen_rollout')
      self._counter += T
      self.total_step_count += T
      self._last_success = res['success'][-1]
      return res
    res = out if out is not None else {}
    sc, self.scalar_api = self.scalar_api, False          # (the step loop stacks batched tensors; with scalar_api step() returns numpy / python scalars)
    try:
      rows = [self.step(a[t].reshape(self.num_envs, self.N_ROBOT)) for t in range(T)]
    finally:
      self.scalar_api = sc
    res['obs'] = torch.stack([r[0] for r in rows]); res['reward'] = torch.stack([r[1] for r in rows]); res['done'] = torch.stack([r[2] for r in rows])
    res['success'] = torch.stack([r[3]['success'] for r in rows]); res['status'] = torch.stack([r[3]['status'] for r in rows])
    return res

  def _get_obs(self):
    with torch.cuda.device(self.device):
      obs = self._observe(noise=True)
    self._counter += 1
    return obs[0].cpu().numpy() if self.scalar_api else obs

  get_obs = _get_obs

  def compute_reward(self, obs):
    """Kitchen.compute_reward (kitchen.py:177-178): like the reference's, it reads the simulator's CURRENT mocap and site positions"""
    o = torch.as_tensor(obs, dtype=torch.float64, device=self.device).reshape(-1, self.OBS_DIM)
    if o.shape[0] != self.num_envs:
      raise ValueError('compute_reward(obs): one observation per env (the reward also reads each env\'s mocap / site positions)')
    r, _ = self._reward(o.contiguous())
    return float(r[0]) if self.scalar_api else r

  def is_successful(self, obs=None):
    if obs is None:                                                               # kitchen.py:180-182: obs = self._get_obs(): a FRESH noisy reading
      with torch.cuda.device(self.device):
        o = self._observe(noise=True)
      self._counter += 1
    else:
      o = torch.as_tensor(obs, dtype=torch.float64, device=self.device).reshape(-1, self.OBS_DIM)
    s = (o[:, 9:23] - o[:, 32:46]).norm(dim=1) <= 0.3                             # kitchen.py:180-183
    return bool(s[0]) if self.scalar_api else s

  # ------------------------------------------------------------------ goals (kitchen.py:106-112)
  def get_next_goal(self):
    return self._goal_states[0]

  def reset_goal(self, goal=None, mask=None):
    g = torch.as_tensor(self.get_next_goal() if goal is None else goal, dtype=torch.float64, device=self.device).expand(self.num_envs, 23)
    if mask is None:
      self.goal_t.copy_(g)
    else:
      m = torch.as_tensor(mask, device=self.device).bool()
      self.goal_t[m] = g[m]

  def get_task(self):
    return self._task

  def get_init_states(self):
    return self._initial_states

  @property
  def goal(self):
    return self.goal_t[0].cpu().numpy() if self.scalar_api else self.goal_t

  def set_state(self, qpos, qvel):
    self.qpos.copy_(torch.as_tensor(qpos, dtype=torch.float64, device=self.device).reshape(self.num_envs, self.NV))
    self.qvel.copy_(torch.as_tensor(qvel, dtype=torch.float64, device=self.device).reshape(self.num_envs, self.NV))

  def state_dict(self):
    keys = ('qpos', 'qvel', 'mocap_pos', 'goal_t', 'last_qp_robot', 'att', 'steps_since_reset', 'interventions', 'fail_count', 'lifelong_return_t',
            'steps_since_goal_change', 'last_obs')
    return {k: getattr(self, k).clone() for k in keys} | {'counter': self._counter, 'total_step_count': self.total_step_count}

  def load_state_dict(self, sd):
    for k, v in sd.items():
      if k == 'counter':
        self._counter = int(v)
      elif k == 'total_step_count':
        self.total_step_count = int(v)
      else:
        getattr(self, k).copy_(v)
