"""earl_benchmark_amd -- MI355X-native drop-in for the step()/reset() hot path of EARL.

Public surface mirrors the reference's loader (reference: earl_benchmark/__init__.py):

    import earl_benchmark_amd as earl_benchmark
    loader = earl_benchmark.EARLEnvs('tabletop_manipulation', reward_type='sparse', num_envs=4096)
    train_env, eval_env = loader.get_envs()
    initial_states, goal_states = loader.get_initial_states(), loader.get_goal_states()
    forward_demos, reverse_demos = loader.get_demonstrations()

Extra keyword arguments of this build: `num_envs` (batch of independent env instances stepped by one HIP
kernel launch), `device`, `seed`, `env_offset` (global id of env 0 when sharding over GPUs), `auto_reset`.
Environment arithmetic runs only in this build's own native code (csrc/): the HIP library on MI355X; there is no CPU fallback.  (The tabletop's per-env
functions also exist as a host build, csrc/libearl_host.so -- used only when a caller asks for device='cpu'.)
"""
import os

import numpy as np

from . import tables

__version__ = '0.1.0'

# reference: earl_benchmark/__init__.py:16-47
deployment_eval_config = {
    'tabletop_manipulation': {'num_initial_state_samples': 1, 'num_goals': 4, 'train_horizon': int(2e5), 'eval_horizon': 200},
    'sawyer_door': {'num_initial_state_samples': 1, 'num_goals': 1, 'train_horizon': int(2e5), 'eval_horizon': 300},
    'sawyer_peg': {'num_initial_state_samples': 15, 'num_goals': 1, 'train_horizon': int(1e5), 'eval_horizon': 200},
    'kitchen': {'num_initial_state_samples': 1, 'train_horizon': int(1e5), 'eval_horizon': 400, 'task': 'all_pairs'},
    'minitaur': {'num_initial_state_samples': 1, 'num_goals': 4, 'train_horizon': int(1e5), 'eval_horizon': 1000},
}
# reference: earl_benchmark/__init__.py:50-81
continuing_eval_config = {
    'tabletop_manipulation': {'num_initial_state_samples': 1, 'num_goals': 4, 'train_horizon': int(5e4), 'goal_change_frequency': 400},
    'sawyer_door': {'num_initial_state_samples': 1, 'num_goals': 1, 'train_horizon': int(5e4), 'goal_change_frequency': 600},
    'sawyer_peg': {'num_initial_state_samples': 15, 'num_goals': 1, 'train_horizon': int(5e4), 'goal_change_frequency': 400},
    'kitchen': {'num_initial_state_samples': 1, 'train_horizon': int(5e4), 'goal_change_frequency': 800, 'task': 'all_pairs'},
    'minitaur': {'num_initial_state_samples': 1, 'num_goals': 4, 'train_horizon': int(1e5), 'goal_change_frequency': 2000},
}

_NOT_BUILT = ('{name}: the reference builds these from env.reset() observations and crashes (`set` of ndarrays, earl_benchmark/__init__.py:213-217); '
              'this build returns nothing for it either.')


class UnpinnedDynamicsWarning(UserWarning):
  """an env whose dynamics could not be checked against the reference's simulator (absent from this build's environment)"""


class EARLEnvs(object):
  """Same constructor and methods as the reference's EARLEnvs (earl_benchmark/__init__.py:83-247)."""

  def __init__(self, env_name, reward_type='sparse', reset_train_env_at_goal=False, setup_as_lifelong_learning=False,
               **kwargs):
    if env_name not in deployment_eval_config:
      raise KeyError(env_name)
    self._env_name = env_name
    self._reward_type = reward_type
    self._reset_train_env_at_goal = reset_train_env_at_goal
    self._setup_as_lifelong_learning = setup_as_lifelong_learning
    self._kwargs = kwargs
    self._batch_kwargs = {k: kwargs[k] for k in ('num_envs', 'device', 'seed', 'env_offset', 'scalar_api', 'auto_reset') if k in kwargs}
    self._info_kwargs = {'info': kwargs['info']} if 'info' in kwargs else {}     # 'full' | 'minimal' info dicts of the physics envs' step() (envs/sawyer_door.py, envs/kitchen.py)

    if env_name == 'kitchen' and reward_type != 'dense':
      raise ValueError('Kitchen environment only supports dense rewards.')  # reference: envs/kitchen.py:91-92 (raised while constructing)
    if not self._setup_as_lifelong_learning:
      cfg = deployment_eval_config[env_name]
      self._train_horizon = kwargs.get('train_horizon', cfg['train_horizon'])
      self._eval_horizon = kwargs.get('eval_horizon', cfg['eval_horizon'])
      self._num_initial_state_samples = kwargs.get('num_initial_state_samples', cfg['num_initial_state_samples'])
    else:
      cfg = continuing_eval_config[env_name]
      self._train_horizon = kwargs.get('train_horizon', cfg['train_horizon'])
      self._num_initial_state_samples = kwargs.get('num_initial_state_samples', cfg['num_initial_state_samples'])
      self._goal_change_frequency = kwargs.get('goal_change_frequency', cfg['goal_change_frequency'])
    # The reference builds its envs here.  This build defers that to the first get_envs(): the tables and demonstrations of EVERY env
    # name (also on machines without a GPU) stay reachable through the real constructor.
    self._envs = None

  def _build_envs(self):
    if self._envs is None:
      if not self._setup_as_lifelong_learning:
        self._envs = (self.get_train_env(), self.get_eval_env())
      else:
        self._envs = (self.get_train_env(lifelong=True), None)
    return self._envs

  @property
  def _train_env(self):
    return self._build_envs()[0]

  @property
  def _eval_env(self):
    return self._build_envs()[1]

  def _make_env(self, reset_at_goal, seed_salt):
    from . import wrappers  # noqa: F401  (imports torch lazily: tables/demos work without a GPU)
    if self._env_name == 'tabletop_manipulation':
      from .envs import tabletop
      kw = dict(self._batch_kwargs)
      kw['seed'] = int(kw.get('seed', 0)) + seed_salt   # train and eval envs draw from different streams
      return tabletop.TabletopManipulation(task_list='rc_r-rc_k-rc_g-rc_b', reward_type=self._reward_type,
                                           reset_at_goal=reset_at_goal,
                                           wide_init_distr=self._kwargs.get('wide_init_distr', False), **kw)
    if self._env_name in ('sawyer_door', 'sawyer_peg', 'kitchen', 'minitaur') and not self._kwargs.get('allow_unpinned_dynamics', False):
      import warnings
      warnings.warn(f'{self._env_name}: the rigid-body dynamics are this build\'s own stepper and contact model; parity with the reference\'s '
                    'MuJoCo 2.1 / PyBullet is UNPINNED (DESIGN.md sections 9-11, 14: Sawyer envs agree with the recorded demonstrations at trajectory '
                    'level, with three constants and the episodes\' start state calibrated on them and checked on held-out episodes (DESIGN.md 16.9); the kitchen has no recordings at all; the '
                    'minitaur\'s robot MODEL is this build\'s own authoring -- the reference ships no URDF).  Pass '
                    'allow_unpinned_dynamics=True to EARLEnvs to silence this.', UnpinnedDynamicsWarning,
                    stacklevel=3)
    if self._env_name == 'sawyer_door':
      # reference: earl_benchmark/__init__.py (sawyer_door.SawyerDoorV2(reward_type=..., reset_at_goal=...)); dynamics: this
      # build's own stepper and contact model -- parity with MuJoCo unpinned
      from .envs import sawyer_door
      kw = dict(self._batch_kwargs)
      kw['seed'] = int(kw.get('seed', 0)) + seed_salt
      return sawyer_door.SawyerDoor(reward_type=self._reward_type, reset_at_goal=reset_at_goal, **kw, **self._info_kwargs)
    if self._env_name == 'sawyer_peg':
      # reference: earl_benchmark/__init__.py:119-122, :146-148 (sawyer_peg.SawyerPegV2(reward_type=..., reset_at_goal=...))
      from .envs import sawyer_peg
      kw = dict(self._batch_kwargs)
      kw['seed'] = int(kw.get('seed', 0)) + seed_salt
      return sawyer_peg.SawyerPeg(reward_type=self._reward_type, reset_at_goal=reset_at_goal, **kw, **self._info_kwargs)
    if self._env_name == 'kitchen':
      # reference: earl_benchmark/__init__.py:133-136, :159-162 (kitchen.Kitchen(task=kitchen_task, reward_type=...)); dynamics: this build's
      # own stepper on the compiled kitchen tables, reduced collision set -- parity with MuJoCo unpinned (envs/kitchen.py)
      from .envs import kitchen
      kw = dict(self._batch_kwargs)
      kw['seed'] = int(kw.get('seed', 0)) + seed_salt
      cfg = continuing_eval_config if self._setup_as_lifelong_learning else deployment_eval_config
      return kitchen.Kitchen(task=self._kwargs.get('kitchen_task', cfg['kitchen']['task']), reward_type=self._reward_type, **kw, **self._info_kwargs)
    if self._env_name == 'minitaur':
      # reference: earl_benchmark/__init__.py:119-125, :164-169 (minitaur_gym_env.GoalConditionedMinitaurBulletEnv(), no arguments); dynamics: this
      # build's own stepper on this build's own robot model -- parity with PyBullet unpinned and model-less (envs/minitaur.py)
      from .envs import minitaur
      kw = dict(self._batch_kwargs)
      kw['seed'] = int(kw.get('seed', 0)) + seed_salt
      return minitaur.Minitaur(**kw)
    raise KeyError(self._env_name)

  def get_train_env(self, lifelong=False):
    from . import wrappers
    train_env = self._make_env(self._reset_train_env_at_goal, seed_salt=0)
    train_env = wrappers.PersistentStateWrapper(train_env, episode_horizon=self._train_horizon)
    if not lifelong:
      return train_env
    return wrappers.LifelongWrapper(train_env, self._goal_change_frequency)

  def get_eval_env(self):
    from . import wrappers
    eval_env = self._make_env(False, seed_salt=0x9E3779B9)
    return wrappers.PersistentStateWrapper(eval_env, episode_horizon=self._eval_horizon)

  def has_demos(self):
    return self._env_name in ['tabletop_manipulation', 'sawyer_door', 'sawyer_peg']

  def get_envs(self):
    if not self._setup_as_lifelong_learning:
      return self._train_env, self._eval_env
    return self._train_env

  def get_initial_states(self, num_samples=None):
    """Always returns initial states of shape N x state_dim (reference :185-219)."""
    del num_samples
    if self._env_name == 'minitaur':
      # the reference builds these from env.reset() observations and crashes (`set` of ndarrays, :213-217)
      raise NotImplementedError(_NOT_BUILT.format(name='minitaur'))
    return tables.initial_states(self._env_name)

  def get_goal_states(self):
    if self._env_name == 'minitaur':
      return None  # the reference falls off the end of the function (:221-236)
    return tables.goal_states(self._env_name)

  def get_demonstrations(self):
    """(forward_demos, reverse_demos): dicts with observations, actions, rewards, terminals, next_observations,
    infos -- the reference's pickle layout (:238-247), stored here as .npz (a demo_data.pkl next to it is also read)."""
    demo_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'demonstrations')
    try:
      return tuple(load_demo(os.path.join(demo_dir, self._env_name, d)) for d in ('forward', 'reverse'))
    except Exception:
      print('please download the demonstrations corresponding to ', self._env_name)


  def get_demonstrations_on_device(self, device='cuda', buffer=None):
    """The demonstrations as device-resident tensors, for seeding a replay buffer without a host round trip per batch:
    -> (forward, reverse), each a dict of `observations` / `next_observations` [N, D] float32, `actions` [N, A] float32, `rewards` [N, 1]
    float32, `terminals` [N, 1] bool -- the reference's layout (:238-247) minus the empty `infos`.  With `buffer` (a dict of preallocated
    tensors with the same keys and at least N_forward + N_reverse rows) the rows are also written into it, forward first, and the number of
    rows written is returned as a third value."""
    import torch
    demos = self.get_demonstrations()
    if demos is None:
      return None
    out = tuple({k: torch.as_tensor(d[k], device=device) for k in DEMO_KEYS if k != 'infos'} for d in demos)
    if buffer is None:
      return out
    n = 0
    for d in out:
      m = d['observations'].shape[0]
      for k, v in d.items():
        buffer[k][n:n + m].copy_(v.to(buffer[k].dtype))
      n += m
    return out + (n,)


DEMO_KEYS = ('observations', 'actions', 'rewards', 'terminals', 'next_observations', 'infos')


def load_demo(directory):
  """Read one demonstration set: demo_data.npz (this build) or demo_data.pkl (the reference's format)."""
  npz = os.path.join(directory, 'demo_data.npz')
  if os.path.exists(npz):
    with np.load(npz) as z:
      return {k: z[k] for k in DEMO_KEYS}
  import pickle
  with open(os.path.join(directory, 'demo_data.pkl'), 'rb') as f:
    demo = pickle.load(f)
  return {k: np.asarray(demo[k]) for k in DEMO_KEYS}
