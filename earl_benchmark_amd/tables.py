"""Initial-state / goal-state tables returned by EARLEnvs.get_initial_states() / get_goal_states().

The numbers are the reference's module constants (reference: envs/tabletop_manipulation.py:11-16,
envs/sawyer_door.py:13-16, envs/sawyer_peg.py:18-58, envs/kitchen.py:28-85 with Kitchen.get_init_states :103-104),
extracted by tests/golden/make_golden.py into tables.npz (numeric data only).
"""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tables.npz')
_cache = None


def _tables():
  global _cache
  if _cache is None:
    with np.load(_PATH) as z:
      _cache = {k: z[k] for k in z.files}
  return _cache


def initial_states(env_name):
  return _tables()[f'{env_name}_initial_states'].copy()


def goal_states(env_name):
  return _tables()[f'{env_name}_goal_states'].copy()


def get(name):
  return _tables()[name].copy()


def kitchen_tasks():
  """task names of Kitchen(task=...) (reference: envs/kitchen.py:57-85, the keys of `initial_states`)"""
  return sorted(k[len('kitchen_task_'):] for k in _tables() if k.startswith('kitchen_task_'))
