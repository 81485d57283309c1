"""Batched counterparts of the reference's gym wrappers.

reference: earl_benchmark/wrappers/persistent_state_wrapper.py, earl_benchmark/wrappers/lifelong_wrapper.py.
The per-step bookkeeping of both wrappers (steps-since-reset, horizon -> done, steps-since-goal-change,
lifelong return, goal switch + re-read of the observation) is fused into the step/rollout kernels; these
classes only configure the env's kernel parameters and keep the reference's attribute surface
(`num_interventions`, `total_steps`, `lifelong_return`, attribute pass-through incl. private names).
"""
import torch


class Wrapper:
  def __init__(self, env):
    self.env = env

  def __getattr__(self, name):  # persistent_state_wrapper.py:47-48 forwards everything, private names included
    if name == 'env':
      raise AttributeError(name)
    return getattr(self.env, name)

  @property
  def unwrapped(self):
    return getattr(self.env, 'unwrapped', self.env)


def _scalar(env, t, cast):
  return cast(t[0]) if env.scalar_api else t


class PersistentStateWrapper(Wrapper):
  """done=True once steps_since_reset >= episode_horizon; counts interventions (resets) and total steps."""

  def __init__(self, env, episode_horizon):
    super().__init__(env)
    self._episode_horizon = int(episode_horizon)
    self.unwrapped._cfg.horizon = min(self._episode_horizon, 2**31 - 1)

  def reset(self, *args, **kwargs):
    return self.env.reset(*args, **kwargs)      # the reset kernel zeroes steps_since_reset and counts the intervention

  def step(self, action, **kwargs):
    return self.env.step(action, **kwargs)

  def rollout(self, actions, **kwargs):
    return self.env.rollout(actions, **kwargs)

  def rollout_episodes(self, actions, **kwargs):
    return self.env.rollout_episodes(actions, **kwargs)

  def is_successful(self, obs=None):
    return self.env.is_successful(obs)

  @property
  def num_interventions(self):
    u = self.unwrapped
    return _scalar(u, u.interventions, int)

  @property
  def total_steps(self):
    return self.unwrapped.total_step_count


class LifelongWrapper(Wrapper):
  """Every goal_change_frequency steps the goal is resampled and the returned observation is rebuilt with the new
  goal; the reward of that step still refers to the old goal (lifelong_wrapper.py:30-44)."""

  def __init__(self, env, goal_change_frequency):
    super().__init__(env)
    self._goal_change_frequency = int(goal_change_frequency)
    self.unwrapped._cfg.goal_change_frequency = self._goal_change_frequency
    u = self.unwrapped
    u.steps_since_goal_change.zero_()
    u.lifelong_return_t.zero_()

  def reset(self, *args, **kwargs):
    return self.env.reset(*args, **kwargs)      # the reset kernel also zeroes steps_since_goal_change

  def step(self, action, **kwargs):
    return self.env.step(action, **kwargs)

  def rollout(self, actions, **kwargs):
    return self.env.rollout(actions, **kwargs)

  @property
  def lifelong_return(self):
    u = self.unwrapped
    return _scalar(u, u.lifelong_return_t, float)

  @property
  def num_interventions(self):
    u = self.unwrapped
    return _scalar(u, u.interventions, int)
