"""Batched, GPU-resident counterpart of the reference's tabletop_manipulation env.

Mirrors `TabletopManipulation` (reference: earl_benchmark/envs/tabletop_manipulation.py) -- same
constructor arguments, same method names (`reset`, `step`, `reset_goal`, `get_next_goal`,
`compute_reward`, `is_successful`, `get_obs`/`_get_obs`, `set_state`), same observation layout -- but
every call acts on `num_envs` independent env instances whose state lives in HBM as torch tensors,
and all arithmetic runs in the hand-written HIP kernels behind include/earl_tabletop.h
(csrc/tabletop.hip).  There is no CPU FALLBACK: constructing an env on the default device without the HIP library
or without a GPU raises.  device='cpu', asked for by name, runs the same per-env functions compiled for the host
(csrc/libearl_host.so, the `_cpu` entry points of include/earl_tabletop.h; BASELINE configs[0]).

Batched conventions: observations `[N, 12] float32`, rewards `[N] float32`, done / success `[N] bool`
torch tensors on the env's device.  With `num_envs == 1` and `scalar_api=True` the env returns what
the reference returns (numpy obs `[12]`, python float reward, python bool done, `{}`), for code
written against the reference.
"""
import contextlib
import ctypes as C

import numpy as np
import torch

from .. import _abi
from ..spaces import Box

INT32_MAX = 2**31 - 1

# reference: tabletop_manipulation.py:11-16
initial_states = np.array([[0.0, 0.0, 2.5, 0.0, -1., -1.]])
goal_states = np.array([[0.0, 0.0, -2.5, -1.0, -1., -1.],
                        [0.0, 0.0, -2.5, 1.0, -1., -1.],
                        [0.0, 0.0, 0.0, 2.0, -1., -1.],
                        [0.0, 0.0, 0.0, -2.0, -1., -1.]])
TARGET_COLORS = ['r', 'g', 'b', 'k']  # :35 -- task 'rc_<colour>' moves the mug to goal_states[index(colour)]


def task_goal_rows(task_list):
  """get_next_goal (:62-76): a task string 'rc_k' -> goal = initial_state with the mug slot set to the target of
  colour k.  Returns the goal-table rows [n_tasks, 6] in task order (uniformly sampled at reset)."""
  rows = []
  for task in task_list.split('-'):
    goal = initial_states[0].copy()
    for sub in task.split('__'):
      obj, colour = sub.split('_')
      if obj != 'rc':
        raise ValueError(f'unknown object {obj!r} in task {task!r} (the model has one mug: "rc")')
      goal[2:4] = goal_states[TARGET_COLORS.index(colour)][2:4]
    rows.append(goal)
  return np.stack(rows)


def _ptr(t):
  return None if t is None else t.data_ptr()


class TabletopManipulation:
  """N independent tabletop envs stepped by one HIP kernel launch (one lane per env)."""

  NOBJ = 1
  OBS_DIM = 12
  NQ = 4
  _FN = 'earl_tabletop_'

  def __init__(self, task_list='rc_r-rc_k-rc_g-rc_b', reward_type='dense', reset_at_goal=False, wide_init_distr=False,
               num_envs=1, device='cuda', seed=0, env_offset=0, scalar_api=None, auto_reset=False):
    dev = torch.device(device)
    if dev.type == 'cpu':
      # asked for by name, never a fallback: the `_cpu` entry points of include/earl_tabletop.h -- this file's kernels' own per-env functions
      # (csrc/tabletop_device.h, tabletop_step.h) compiled for the host (csrc/libearl_host.so), host tensors, no stream.  BASELINE configs[0].
      self._lib = _abi.load_host()
    elif dev.type != 'cuda':
      raise _abi.EarlHipError(f'device={device!r}: the tabletop hot path runs on MI355X (device="cuda"), or on the host when asked for device="cpu"')
    else:
      self._lib = _abi.load()
      if not torch.cuda.is_available():
        raise _abi.EarlHipError('no HIP device visible (torch.cuda.is_available() is False)')
      if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if reward_type not in _abi.REWARD_TYPES:
      raise ValueError(f'reward_type must be sparse|dense, got {reward_type!r}')
    self.device = dev
    self.num_envs = int(num_envs)
    self.scalar_api = (self.num_envs == 1) if scalar_api is None else bool(scalar_api)
    if self.scalar_api and self.num_envs != 1:
      raise ValueError('scalar_api needs num_envs == 1')
    self._task_list = task_list
    self._reward_type = reward_type
    self._reset_at_goal = bool(reset_at_goal)
    self._wide_init_distr = bool(wide_init_distr)
    self.threshold = 0.4
    self.move_distance = 0.2
    self.initial_state = self._initial_states()[0].copy()
    self._goal_list = self._goal_states().copy()

    n = self.num_envs
    self._base_goals = self._task_rows(task_list)                    # rows the RNG samples from
    self._n_sample_goals = len(self._base_goals)
    self._goal_width = self._base_goals.shape[1]
    kw = dict(device=dev)
    self.qpos = torch.zeros(n, self.NQ, dtype=torch.float64, **kw)
    self.attached = torch.full((n,), -1, dtype=torch.int8, **kw)
    self.goal_idx = torch.zeros(n, dtype=torch.int32, **kw)
    self.goal_table = torch.tensor(self._base_goals, dtype=torch.float64, **kw)
    self.steps_since_reset = torch.zeros(n, dtype=torch.int32, **kw)
    self.interventions = torch.zeros(n, dtype=torch.int32, **kw)
    self.steps_since_goal_change = torch.zeros(n, dtype=torch.int32, **kw)
    self.lifelong_return_t = torch.zeros(n, dtype=torch.float64, **kw)
    self.total_step_count = 0
    self._last_success = None

    self._cfg = _abi.TabletopCfg(n=n, env_offset=int(env_offset), reward_type=_abi.REWARD_TYPES[reward_type],
                                 wide_init=int(self._wide_init_distr), reset_at_goal=int(self._reset_at_goal),
                                 horizon=INT32_MAX, goal_change_frequency=0, auto_reset=int(bool(auto_reset)),
                                 n_goals=len(self._base_goals), n_sample_goals=self._n_sample_goals,
                                 seed=int(seed) & (2**64 - 1), counter=0)
    self._st = _abi.TabletopState()
    self._cfg_ref, self._st_ref = C.byref(self._cfg), C.byref(self._st)   # reused by the hot calls
    self._sync_state_ptrs()

    self.action_space = Box(-1.0, 1.0, (3,), np.float32)           # three ctrlrange=[-1,1] motors of the model
    self.observation_space = Box(-np.inf, np.inf, (self.OBS_DIM,), np.float32)
    # like gym's MujocoEnv constructor, leave the env in a reset state
    with self._ctx():
      self._reset_kernel(None, None, want_obs=False)
    self.interventions.zero_()

  # ------------------------------------------------------------------ tables (overridden by the 3obj variant)
  @staticmethod
  def _initial_states():
    return initial_states

  @staticmethod
  def _goal_states():
    return goal_states

  @staticmethod
  def _task_rows(task_list):
    return task_goal_rows(task_list)

  @property
  def unwrapped(self):
    return self

  # ------------------------------------------------------------------ plumbing
  def _sync_state_ptrs(self):
    s = self._st
    s.qpos, s.attached, s.goal_idx, s.goal_table = _ptr(self.qpos), _ptr(self.attached), _ptr(self.goal_idx), _ptr(self.goal_table)
    s.steps_since_reset, s.num_interventions = _ptr(self.steps_since_reset), _ptr(self.interventions)
    s.steps_since_goal_change, s.lifelong_return = _ptr(self.steps_since_goal_change), _ptr(self.lifelong_return_t)
    self._cfg.n_goals = self.goal_table.shape[0]

  def _stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == 'cuda' else None

  def _ctx(self):
    """launches go to the current device's runtime context (nothing to select on the host)"""
    return torch.cuda.device(self.device) if self.device.type == 'cuda' else contextlib.nullcontext()

  def _check(self, rc, what):
    if rc:
      _abi.check(rc, what, self._lib)

  def _next_counter(self, k=1):
    c = self._cfg.counter
    self._cfg.counter = c + k
    return c

  def _as_i32(self, x):
    if x is None:
      return None
    return torch.as_tensor(x, device=self.device).to(torch.int32).contiguous()

  def _reset_kernel(self, mask, next_goal_idx, want_obs=True):
    obs = torch.empty(self.num_envs, self.OBS_DIM, dtype=torch.float32, device=self.device) if want_obs else None
    m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
    g = self._as_i32(next_goal_idx)
    if self.NOBJ == 1:
      rc = self._lib.earl_tabletop_reset(C.byref(self._cfg), C.byref(self._st), _ptr(m), _ptr(g), _ptr(obs), self._stream())
    else:
      rc = self._lib.earl_tabletop3_reset(C.byref(self._cfg), C.byref(self._st), _ptr(m), _ptr(obs), self._stream())
    self._check(rc, 'reset')
    self._next_counter()
    return obs

  def _new_out(self, lead):
    kw = dict(device=self.device)
    obs = torch.empty(*lead, self.OBS_DIM, dtype=torch.float32, **kw)
    rew = torch.empty(*lead, dtype=torch.float32, **kw)
    done = torch.empty(*lead, dtype=torch.bool, **kw)
    succ = torch.empty(*lead, dtype=torch.bool, **kw)
    return (obs, rew, done, succ), _abi.TabletopOut(_ptr(obs), _ptr(rew), _ptr(done), _ptr(succ))

  def _actions(self, action, lead):
    a = torch.as_tensor(action, device=self.device)
    if a.dtype != torch.float32:
      a = a.to(torch.float32)
    a = a.reshape(*lead, 3)
    return a if a.is_contiguous() else a.contiguous()

  # ------------------------------------------------------------------ gym-style API
  def reset(self, mask=None, goal_idx=None):
    """reset() of the reference for every env (or the envs selected by the bool tensor `mask`).  `goal_idx`
    (optional, [N]) injects the goal-table rows instead of sampling, like `reset_goal(goal)` in the reference."""
    with self._ctx():
      obs = self._reset_kernel(mask, goal_idx)
    return obs[0].cpu().numpy() if self.scalar_api else obs

  def _out_struct(self, outs, lead):
    """validate caller-provided output tensors (obs, reward, done, success) and wrap their pointers"""
    want = [(lead + (self.OBS_DIM,), torch.float32), (lead, torch.float32), (lead, torch.bool), (lead, torch.bool)]
    if len(outs) != 4:
      raise ValueError('out must be (obs, reward, done, success)')
    for t, (shape, dt) in zip(outs, want):
      if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
        raise ValueError(f'out tensor must be contiguous {dt} {shape} on {self.device}')
    return _abi.TabletopOut(*(t.data_ptr() for t in outs))

  def step(self, action, next_goal_idx=None, out=None):
    """step(action) of the wrapped reference env for every env: action [N, 3] -> (obs [N, D], reward [N], done [N],
    {'success': [N]}).  Fresh output tensors per call like the reference (callers may keep them); pass
    `out=(obs, reward, done, success)` to write into preallocated tensors instead."""
    n = self.num_envs
    if (torch.is_tensor(action) and action.dtype == torch.float32 and action.device == self.device
        and action.shape == (n, 3) and action.is_contiguous()):
      act = action
    else:
      act = self._actions(action, (n,))
    if out is None:
      outs, ostruct = self._new_out((n,))
    else:
      outs, ostruct = tuple(out), self._out_struct(tuple(out), (n,))
    r64 = None
    if self.scalar_api:        # the reference returns the reward as a Python float (float64), not rounded to float32
      r64 = torch.empty(n, dtype=torch.float64, device=self.device)
      ostruct.reward_f64 = r64.data_ptr()
    ctx = None
    if self.device.type == 'cuda' and torch.cuda.current_device() != self.device.index:   # launches go to the current device's runtime context
      ctx = torch.cuda.device(self.device)
      ctx.__enter__()
    try:
      if self.NOBJ == 1:
        g = self._as_i32(next_goal_idx)
        rc = self._lib.earl_tabletop_step(self._cfg_ref, self._st_ref, act.data_ptr(), _ptr(g), C.byref(ostruct), self._stream())
      else:
        rc = self._lib.earl_tabletop3_step(self._cfg_ref, self._st_ref, act.data_ptr(), C.byref(ostruct), self._stream())
    finally:
      if ctx is not None:
        ctx.__exit__(None, None, None)
    if rc:
      self._check(rc, 'step')
    self._cfg.counter += 1          # every call uses the current Philox counter, then advances it
    self.total_step_count += 1
    obs, rew, done, succ = outs
    self._last_success = succ
    if self.scalar_api:
      return obs[0].cpu().numpy(), float(r64[0]), bool(done[0]), {}
    return obs, rew, done, {'success': succ}

  def rollout_episodes(self, actions, episodes=None, out=None):
    """`episodes` evaluation episodes of every env, back to back (each = reset() + T steps), in ONE kernel launch when the fused kernel
    applies (include/earl_tabletop.h: earl_tabletop_eval_episodes).  actions [T, N, 3] (replayed by every episode; give `episodes`) or
    [E, T, N, 3].  -> (obs [E,T,N,D], reward [E,T,N], done [E,T,N], success [E,T,N]); bit-identical to E calls of
    rollout(actions, reset_first=True)."""
    if self.NOBJ != 1:
      raise NotImplementedError('rollout_episodes: single-object env only')
    with self._ctx():
      a = torch.as_tensor(actions, device=self.device)
      if a.dim() == 4:
        E, T = int(a.shape[0]), int(a.shape[1])
        if episodes is not None and int(episodes) != E:
          raise ValueError(f'episodes = {episodes} but actions hold {E}')
        act, stride = self._actions(a, (E, T, self.num_envs)), T * self.num_envs * 3
      else:
        if episodes is None:
          raise ValueError('rollout_episodes: actions [T, N, 3] are replayed by every episode -- say how many (episodes=E), or pass [E, T, N, 3]')
        E, T = int(episodes), int(a.shape[0])
        act, stride = self._actions(a, (T, self.num_envs)), 0
      if out is None:
        outs, ostruct = self._new_out((E, T, self.num_envs))
      else:
        outs = tuple(out)
        ostruct = self._out_struct(outs, (E, T, self.num_envs))
      rc = self._lib.earl_tabletop_eval_episodes(self._cfg_ref, self._st_ref, E, T, act.data_ptr(), stride, C.byref(ostruct), self._stream())
    self._check(rc, 'eval_episodes')
    self._cfg.counter += E * (T + 1)
    self.total_step_count += E * T
    self._last_success = outs[3][-1, -1]
    return outs

  def make_step_graph(self, T, policy=None):
    """Closed-loop stepping without the per-call host cost: a captured HIP graph of T step launches (see `StepGraph`)."""
    return StepGraph(self, T, policy)

  def rollout(self, actions, out=None, reset_first=False):
    """T steps in one kernel launch: actions [T, N, 3] -> (obs [T,N,D], reward [T,N], done [T,N], success [T,N]).
    Bit-identical to T calls of step().  `out`: optional tuple of preallocated output tensors to write into.
    `reset_first=True` folds a reset() of every env into the same launch (one evaluation episode per call)."""
    with self._ctx():
      a = torch.as_tensor(actions, device=self.device)
      T = a.shape[0]
      act = self._actions(a, (T, self.num_envs))
      if out is None:
        outs, out = self._new_out((T, self.num_envs))
      else:
        outs = tuple(out)
        out = self._out_struct(outs, (T, self.num_envs))
      if reset_first and self.NOBJ == 1:
        fn = self._lib.earl_tabletop_reset_rollout
      else:
        if reset_first:
          self._reset_kernel(None, None, want_obs=False)
        fn = self._lib.earl_tabletop_rollout if self.NOBJ == 1 else self._lib.earl_tabletop3_rollout
      rc = fn(C.byref(self._cfg), C.byref(self._st), T, act.data_ptr(), C.byref(out), self._stream())
    self._check(rc, 'rollout')
    self._cfg.counter += T + (1 if reset_first and self.NOBJ == 1 else 0)   # step t used counter (+1 after a fused reset) + t
    self.total_step_count += T
    self._last_success = outs[3][-1]
    return outs

  def _observe(self, want=('obs',)):
    kw = dict(device=self.device)
    n = self.num_envs
    obs = torch.empty(n, self.OBS_DIM, dtype=torch.float32, **kw) if 'obs' in want else None
    rew = torch.empty(n, dtype=torch.float32, **kw) if 'reward' in want else None
    succ = torch.empty(n, dtype=torch.bool, **kw) if 'success' in want else None
    if self.NOBJ == 1:
      out = _abi.TabletopOut(_ptr(obs), _ptr(rew), None, _ptr(succ))
      with self._ctx():
        self._check(self._lib.earl_tabletop_observe(C.byref(self._cfg), C.byref(self._st), C.byref(out), self._stream()), 'observe')
    else:
      # 3obj: obs through a no-op masked reset, reward/success through the pure reward kernel
      with self._ctx():
        o = torch.empty(n, self.OBS_DIM, dtype=torch.float32, **kw)
        zero = torch.zeros(n, dtype=torch.uint8, **kw)
        self._check(self._lib.earl_tabletop3_reset(C.byref(self._cfg), C.byref(self._st), zero.data_ptr(), o.data_ptr(), self._stream()), 'observe')
        if rew is not None or succ is not None:
          self._check(self._lib.earl_tabletop3_reward(n, o.data_ptr(), self._cfg.reward_type, _ptr(rew), _ptr(succ), self._stream()), 'reward')
        obs = o if obs is not None else None
    return obs, rew, succ

  def _get_obs(self):
    obs = self._observe(('obs',))[0]
    return obs[0].cpu().numpy() if self.scalar_api else obs

  def get_obs(self):
    return self._get_obs()

  def compute_reward(self, obs):
    """compute_reward(obs) of the reference on an observation batch [M, D] (any M)."""
    r, _ = self._reward(obs, want_reward=True, want_success=False)
    return float(r[0]) if self.scalar_api and r.numel() == 1 else r

  def is_successful(self, obs=None):
    if obs is None:
      s = self._observe(('success',))[2]
    else:
      _, s = self._reward(obs, want_reward=False, want_success=True)
    return bool(s[0]) if self.scalar_api and s.numel() == 1 else s

  def _reward(self, obs, want_reward, want_success):
    o = torch.as_tensor(obs, device=self.device).to(torch.float32).reshape(-1, self.OBS_DIM).contiguous()
    m = o.shape[0]
    r = torch.empty(m, dtype=torch.float32, device=self.device) if want_reward else None
    s = torch.empty(m, dtype=torch.bool, device=self.device) if want_success else None
    with self._ctx():
      if self.NOBJ == 1:
        rc = self._lib.earl_tabletop_reward(m, o.data_ptr(), self._cfg.reward_type, self._cfg.wide_init, _ptr(r), _ptr(s), self._stream())
      else:
        rc = self._lib.earl_tabletop3_reward(m, o.data_ptr(), self._cfg.reward_type, _ptr(r), _ptr(s), self._stream())
    self._check(rc, 'reward')
    return r, s

  # ------------------------------------------------------------------ goals / state injection
  def get_next_goal(self):
    """Sample a goal per env (uniform over the task list, Philox keyed by global env id) WITHOUT setting it."""
    idx = self._sample_goal_rows()
    g = self.goal_table[idx.long()]
    return g[0].cpu().numpy() if self.scalar_api else g

  def _sample_goal_rows(self):
    # same draw layout as the kernels: block(draw=0)[0] * n_sample >> 32; done through a masked-out reset on a scratch copy
    scratch = TabletopStateScratch(self)
    return scratch.sample()

  def reset_goal(self, goal=None, mask=None):
    """reset_goal(goal=None) (:78-81).  goal None -> resample; a [D]-vector -> every (masked) env; [N, D] -> per env."""
    sel = slice(None) if mask is None else torch.as_tensor(mask, device=self.device).bool()
    if goal is None:
      idx = self._sample_goal_rows()
      self.goal_idx[sel] = idx[sel]
      return
    g = torch.as_tensor(np.asarray(goal) if not torch.is_tensor(goal) else goal, device=self.device).to(torch.float64)
    if g.ndim == 1:
      g = g.expand(self.num_envs, -1)
    if g.shape != (self.num_envs, self._goal_width):
      raise ValueError(f'goal must have shape [{self._goal_width}] or [{self.num_envs}, {self._goal_width}]')
    # custom goals live in per-env rows appended after the sampled rows
    base = self._n_sample_goals
    if self.goal_table.shape[0] != base + self.num_envs:
      cur = self.goal_table[self.goal_idx.long()]
      self.goal_table = torch.cat([self.goal_table[:base], cur], 0).contiguous()
      self.goal_idx.copy_(torch.arange(base, base + self.num_envs, device=self.device, dtype=torch.int32))
      self._sync_state_ptrs()
    rows = torch.arange(base, base + self.num_envs, device=self.device)
    self.goal_table[rows[sel]] = g[sel]
    self.goal_idx[sel] = rows[sel].to(torch.int32)

  @property
  def goal(self):
    g = self.goal_table[self.goal_idx.long()]
    return g[0].cpu().numpy() if self.scalar_api else g

  @property
  def attached_object(self):
    """(-1,-1) free / (0,0) holding, like the reference (scalar API); the int8 tensor otherwise."""
    if self.scalar_api:
      k = int(self.attached[0])
      return (-1, -1) if k < 0 else (0.5 * k, 0.5 * k) if self.NOBJ > 1 else (0, 0)
    return self.attached

  def set_state(self, qpos, qvel=None):
    """set_state(qpos) (:83-87): overwrite the gripper/mug coordinates (only the first NQ entries are used)."""
    del qvel
    q = torch.as_tensor(np.asarray(qpos) if not torch.is_tensor(qpos) else qpos, device=self.device).to(torch.float64)
    q = q.reshape(-1, q.shape[-1])[:, :self.NQ]
    self.qpos.copy_(q.expand(self.num_envs, -1))

  # ------------------------------------------------------------------ checkpoint / counters
  def state_dict(self):
    return {'qpos': self.qpos.clone(), 'attached': self.attached.clone(), 'goal_idx': self.goal_idx.clone(),
            'goal_table': self.goal_table.clone(), 'steps_since_reset': self.steps_since_reset.clone(),
            'interventions': self.interventions.clone(), 'steps_since_goal_change': self.steps_since_goal_change.clone(),
            'lifelong_return': self.lifelong_return_t.clone(), 'total_step_count': self.total_step_count,
            'rng_counter': int(self._cfg.counter), 'seed': int(self._cfg.seed)}

  def load_state_dict(self, sd):
    self.goal_table = sd['goal_table'].to(self.device).clone().contiguous()
    for name, key in (('qpos', 'qpos'), ('attached', 'attached'), ('goal_idx', 'goal_idx'),
                      ('steps_since_reset', 'steps_since_reset'), ('interventions', 'interventions'),
                      ('steps_since_goal_change', 'steps_since_goal_change'), ('lifelong_return_t', 'lifelong_return')):
      getattr(self, name).copy_(sd[key])
    self.total_step_count = int(sd['total_step_count'])
    self._cfg.counter = int(sd['rng_counter'])
    self._cfg.seed = int(sd['seed'])
    self._sync_state_ptrs()


class TabletopStateScratch:
  """Draws goal rows with the reset kernel on throw-away state (keeps get_next_goal() side-effect free)."""

  def __init__(self, env):
    self.env = env

  def sample(self):
    e = self.env
    n, dev = e.num_envs, e.device
    st = _abi.TabletopState()
    qpos = torch.empty(n, e.NQ, dtype=torch.float64, device=dev)
    att = torch.empty(n, dtype=torch.int8, device=dev)
    gi = torch.empty(n, dtype=torch.int32, device=dev)
    steps = torch.empty(n, dtype=torch.int32, device=dev)
    iv = torch.zeros(n, dtype=torch.int32, device=dev)
    sgc = torch.empty(n, dtype=torch.int32, device=dev)
    lr = torch.empty(n, dtype=torch.float64, device=dev)
    st.qpos, st.attached, st.goal_idx, st.goal_table = qpos.data_ptr(), att.data_ptr(), gi.data_ptr(), e.goal_table.data_ptr()
    st.steps_since_reset, st.num_interventions = steps.data_ptr(), iv.data_ptr()
    st.steps_since_goal_change, st.lifelong_return = sgc.data_ptr(), lr.data_ptr()
    cfg = _abi.TabletopCfg.from_buffer_copy(e._cfg)
    cfg.wide_init, cfg.reset_at_goal = 0, 0
    with e._ctx():
      if e.NOBJ == 1:
        rc = e._lib.earl_tabletop_reset(C.byref(cfg), C.byref(st), None, None, None, e._stream())
      else:
        rc = e._lib.earl_tabletop3_reset(C.byref(cfg), C.byref(st), None, None, e._stream())
    e._check(rc, 'sample_goal')
    e._next_counter()
    return gi


class StepGraph:
  """T gym-style `step()` launches of one env batch captured once into a HIP graph and replayed with a single host call: the closed-loop
  counterpart of `rollout()` (which needs all T actions up front).  One replay costs one graph launch instead of T x (ctypes call +
  argument marshalling + kernel launch): at N = 4096 the per-step wall time drops from ~14 us to about the kernel's own ~4 us.

    g = env.make_step_graph(T)                  # action ring: the caller (or its own captured kernels) fills g.actions[t] before replay
    g = env.make_step_graph(T, policy=pi)       # pi(obs [N, D]) -> actions [N, 3] is captured INTO the graph between the steps: step t
                                                #   consumes pi(observation of step t - 1); `g.obs_in` holds the observation the first
                                                #   step of a replay starts from and is refreshed by the graph's last node
    g.replay()                                  # obs / reward / done / success of the T steps are in g.obs[t], g.reward[t], ...

  Each captured step is exactly `step()` (same kernel, same state tensors), so a replay is bit-identical to T eager calls -- also with lifelong goal
  switching (the reference's train-env loop, wrappers/lifelong_wrapper.py:30-44) and auto-reset, whose draws use the launch's Philox counter: a kernel
  ARGUMENT would be frozen at capture time, so the captured launch of step t carries the OFFSET t and adds a base it reads from one device word
  (earl_tabletop_state.counter_base), which replay() refreshes from the env's counter before every replay."""

  def __init__(self, env, T, policy=None):
    u = env.unwrapped if hasattr(env, 'unwrapped') else env
    if u.scalar_api:
      raise ValueError('make_step_graph is for the batched API (scalar_api=False)')
    if u.device.type != 'cuda':
      raise ValueError('make_step_graph captures HIP launches: device="cuda" only')
    self.env, self.T, self.policy = u, int(T), policy
    n, dev = u.num_envs, u.device
    with torch.cuda.device(dev):
      self.actions = torch.zeros(self.T, n, 3, dtype=torch.float32, device=dev)
      (self.obs, self.reward, self.done, self.success), _ = u._new_out((self.T, n))
      self.obs_in = u._observe(('obs',))[0].clone()
      structs = [_abi.TabletopOut(self.obs[t].data_ptr(), self.reward[t].data_ptr(), self.done[t].data_ptr(), self.success[t].data_ptr())
                 for t in range(self.T)]
      step_fn = u._lib.earl_tabletop_step if u.NOBJ == 1 else u._lib.earl_tabletop3_step
      self.counter_dev = torch.zeros(1, dtype=torch.int64, device=dev)      # the Philox counter of the replay's first step (two's complement of the uint64)
      host_counter = int(u._cfg.counter)

      def launch(t):
        u._cfg.counter = t                                     # the captured launch draws with *counter_base + t
        if u.NOBJ == 1:
          rc = step_fn(u._cfg_ref, u._st_ref, self.actions[t].data_ptr(), None, C.byref(structs[t]), u._stream())
        else:
          rc = step_fn(u._cfg_ref, u._st_ref, self.actions[t].data_ptr(), C.byref(structs[t]), u._stream())
        if rc:
          _abi.check(rc, 'step (graph capture)')
      if policy is not None:                                   # warm the policy up outside the capture (lazy library initialisation)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
          policy(self.obs_in)
        torch.cuda.current_stream(dev).wait_stream(side)
      self.graph = torch.cuda.CUDAGraph()
      u._st.counter_base = self.counter_dev.data_ptr()
      try:
        with torch.cuda.graph(self.graph):
          prev = self.obs_in
          for t in range(self.T):
            if policy is not None:
              self.actions[t].copy_(policy(prev).to(torch.float32).reshape(n, 3))
            launch(t)
            prev = self.obs[t]
          if policy is not None:
            self.obs_in.copy_(self.obs[self.T - 1])
      finally:                                                 # eager calls keep the counter as their argument
        u._st.counter_base = None
        u._cfg.counter = host_counter

  def replay(self):
    """run the T captured steps (asynchronous, on torch's current stream); -> (obs, reward, done, {'success': success}), each [T, N, ...]"""
    u = self.env
    c = int(u._cfg.counter)
    self.counter_dev.fill_(c - (1 << 64) if c >= (1 << 63) else c)
    self.graph.replay()
    u._cfg.counter += self.T
    u.total_step_count += self.T
    u._last_success = self.success[-1]
    return self.obs, self.reward, self.done, {'success': self.success}
