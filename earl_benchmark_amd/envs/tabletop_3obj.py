"""Batched counterpart of the reference's 3-object tabletop variant
(reference: earl_benchmark/envs/tabletop_manipulation_3obj.py -- not wired into the reference's loader).

Same kernels, NOBJ = 3: qpos [N, 8] (gripper + three objects), the closest object inside the 0.4 grasp
radius is latched (:99-112), `attached` holds the object index -1/0/1/2 (the reference's
(-1,-1)/(0,0)/(.5,.5)/(1,1) flag pairs), obs [N, 20], success radius 0.4 on the 8-vector (:153-159).
"""
import numpy as np

from .tabletop import TabletopManipulation as _Base

# reference: tabletop_manipulation_3obj.py:11-17
initial_states = np.array([[0.0, 0.0, 2.5, 0.0, 2.5, -1.0, 2.5, 1.0, -1., -1.]])
goal_states = np.array([[0.0, 0.0, 0.0, -2.0, 0.0, 2.0, -2.5, 1.0, -1., -1.]])


class TabletopManipulation(_Base):
  NOBJ = 3
  OBS_DIM = 20
  NQ = 8

  def __init__(self, reward_type='dense', reset_at_goal=False, num_envs=1, device='cuda', seed=0, env_offset=0,
               scalar_api=None, auto_reset=False):
    super().__init__(task_list='', reward_type=reward_type, reset_at_goal=reset_at_goal, wide_init_distr=False,
                     num_envs=num_envs, device=device, seed=seed, env_offset=env_offset, scalar_api=scalar_api,
                     auto_reset=auto_reset)

  @staticmethod
  def _initial_states():
    return initial_states

  @staticmethod
  def _goal_states():
    return goal_states

  @staticmethod
  def _task_rows(task_list):
    del task_list
    return goal_states.copy()  # get_next_goal: np.random.randint over the goal list (:52-56)
