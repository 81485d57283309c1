"""Batched, GPU-resident counterpart of the reference's Sawyer peg-insertion env (BASELINE config 3, SURVEY.md 8 rows a11, a12,
a14, a15).

Mirrors `SawyerPegV2` (reference: earl_benchmark/envs/sawyer_peg.py) -- constructor arguments, `reset`, `step`, `reset_goal`,
`get_next_goal`, `compute_reward` (sparse), `is_successful`, `_get_obs`, observation layout (hand xyz, gripper opening, pegHead xyz,
goal[7]) -- for `num_envs` independent instances stepped by ONE kernel launch (csrc/physics.hip, nv = 15: 7 arm hinges, 2 claw slides,
the peg's free joint as three translations + a quaternion).

STATUS: the dynamics are this build's own stepper and contact model (the peg as a chain of inscribed spheres against the
gripper plates and the hole block, its corners as points against the table and the block); parity with MuJoCo is UNPINNED
(DESIGN.md section 9).  Pinned by the reference's data: the sparse success rule (bit-exact on the 1,815 demonstration rows), the
reset observation (hand pose, gripper opening 1.0, pegHead at z = 0.02), initial / goal tables.  The dense reward is evaluated in the
kernel from metaworld's reward_utils / _gripper_caging_reward as restated in oracle/sawyer_oracle.py -- upstream code absent from the
reference tree, hence UNPINNED (the kernel is tested against that restatement only).
"""
import numpy as np
import torch

from .sawyer_door import SawyerDoor

# reference: sawyer_peg.py:18-58
initial_states = np.array([[0.00615235, 0.6001898, 0.19430117, 1.0, x, y, 0.02] for x, y in (
    (0.00313463, 0.68326396), (-0.04035005, 0.67949003), (0.02531051, 0.6074387), (0.05957219, 0.6271171), (-0.07566337, 0.62575287),
    (-0.01177235, 0.55206996), (0.02779735, 0.54707706), (0.01835314, 0.5329686), (0.02690855, 0.6263067), (0.01766127, 0.59630984),
    (0.0560186, 0.6634998), (-0.03950658, 0.6323736), (-0.03216827, 0.5247563), (0.01265727, 0.69466716), (0.05076993, 0.6025737))])
goal_states = np.array([[0.0, 0.6, 0.2, 1.0, -0.3 + 0.03, 0.6, 0.0 + 0.13]])
# reference: sawyer_peg.py:55-60 (peg positions only)
wide_initial_states = np.array([[-0.3, 0.8, 0.02], [-0.4, 0.8, 0.02], [-0.3, 0.9, 0.02], [-0.4, 0.9, 0.02], [-0.2, 0.8, 0.02], [-0.2, 0.75, 0.02],
                                [-0.2, 0.9, 0.02], [-0.1, 0.77, 0.02], [0.0, 0.9, 0.02], [0.1, 0.8, 0.02], [0.15, 0.75, 0.02], [-0.3, 0.4, 0.02],
                                [-0.4, 0.4, 0.02], [-0.3, 0.45, 0.02], [-0.4, 0.45, 0.02], [-0.2, 0.4, 0.02], [-0.2, 0.45, 0.02], [-0.2, 0.38, 0.02],
                                [-0.1, 0.42, 0.02], [0.0, 0.45, 0.02], [0.1, 0.36, 0.02], [0.15, 0.44, 0.02]])


class SawyerPeg(SawyerDoor):
  """N independent Sawyer peg-insertion envs; state (qpos [N,16], qvel [N,15], mocap) lives in HBM."""

  MODEL = 'sawyer_peg'
  RECORDED_HAND_INIT = (0.0, 0.6, 0.2)       # sawyer_peg.py:70: both variants reset the hand here
  TARGET_RADIUS = 0.05                    # sawyer_peg.py:62

  def __init__(self, reward_type='sparse', reset_at_goal=False, wide_init=False, **kw):
    self.wide_init = bool(wide_init)
    super().__init__(reward_type=reward_type, reset_at_goal=reset_at_goal, **kw)

  def _task_constants(self):
    # sawyer_peg.py:79-92
    self.obj_init_pos = np.array([0, 0.6, 0.02])
    self.hand_init_pos = np.array([0, 0.6, 0.2])
    self.initial_states = initial_states.copy()
    self.goal_states = goal_states.copy()

  def _after_settle(self):
    # SawyerXYZEnv._reset_hand ends with self.init_tcp = self.tcp_center [UPSTREAM]: midpoint of the two finger sites at the settled pose
    q, v = self._reset_state
    kw = dict(dtype=torch.float64, device=self.device)
    _, _, att = self.model.forward(q[None].contiguous(), v[None].contiguous(), torch.tensor([[float(x) for x in self.hand_init_pos]], **kw),
                                   torch.tensor([[1.0, 0.0, 1.0, 0.0]], **kw), torch.tensor([[-1.0, 1.0]], **kw))
    names = self.model.att_names
    tcp = 0.5 * (att[0, names.index('rightEndEffector')] + att[0, names.index('leftEndEffector')])
    self._cfg.init_tcp[:] = [float(x) for x in tcp.cpu()]

  def compute_reward(self, obs, actions=None):
    if self._reward_type != 'sparse':
      raise NotImplementedError('sawyer_peg: the dense reward reads simulator state (sites, pads, reset-time positions) and is evaluated '
                                'inside step / rollout only; compute_reward(obs) is available for the sparse reward')
    return super().compute_reward(obs, actions)

  def _task_cfg(self, cfg, names):
    cfg.att_obj = names.index('pegHead')                    # _get_pos_objects, sawyer_peg.py:186-187
    # dense reward (sawyer_peg.py:231-299; metaworld's reward_utils / _gripper_caging_reward restated [UPSTREAM, unpinned])
    cfg.att_grasp, cfg.att_lpad, cfg.att_rpad = names.index('pegGrasp'), names.index('leftpad'), names.index('rightpad')
    for k, c in enumerate(self.model.tables['peg_box_corners']):
      cfg.box_corners[k][:] = [float(x) for x in c]
    cfg.obj_dof, cfg.obj_kind = int(self.model.struct.ball_dof) - 3, 1
    cfg.success_radius = self.TARGET_RADIUS
    pos_box = goal_states[0][4:] - np.array([0.03, 0.0, 0.13])          # :196
    if not self._reset_at_goal:
      # random_init [UPSTREAM default; the 15 recorded initial states and every forward demonstration start from it]:
      # pos_peg = first half of U(_random_reset_space) = U(obj_low, obj_high) (:65-66, :98-101, :210-212), redrawn while
      # within 0.1 of the hole block in xy
      cfg.obj_low[:] = (0.0, 0.5, 0.02)
      cfg.obj_high[:] = (0.2, 0.7, 0.02)
      cfg.obj_reject_xy[:] = [float(pos_box[0]), float(pos_box[1])]
      cfg.obj_reject_radius = 0.1
      # get_next_goal without reset_at_goal: a row of goal_states (:144-148); reset_model -> reset_goal() restores it on every reset (:195)
      self._goal_table = torch.tensor(goal_states, dtype=torch.float64, device=self.device).contiguous()
      cfg.n_goal_rows, cfg.goal_table = len(goal_states), self._goal_table.data_ptr()
      if self.wide_init:
        # :200-209: with probability 1/2 the draw above, otherwise a row of wide_initial_states - (-0.1, 0, 0) + U(-0.02, 0.02)^3
        self._wide_table = torch.tensor(wide_initial_states, dtype=torch.float64, device=self.device).contiguous()
        cfg.obj_kind, cfg.n_wide, cfg.wide_table = 2, len(wide_initial_states), self._wide_table.data_ptr()
        cfg.wide_shift[:] = (0.1, 0.0, 0.0)
        cfg.wide_noise = 0.02
    else:
      # :216-227: the peg starts in the hole, goal_pos + U(-0.02, 0.02)^3 with goal_pos = goal - (-0.1, 0, 0); the goal is one of
      # the initial states (get_next_goal :149-152)
      goal_pos = goal_states[0][4:] - np.array([-0.1, 0.0, 0.0])
      cfg.obj_low[:] = [float(x) for x in goal_pos - 0.02]
      cfg.obj_high[:] = [float(x) for x in goal_pos + 0.02]
      cfg.obj_reject_radius = 0.0
      self._goal_table = torch.tensor(initial_states, dtype=torch.float64, device=self.device).contiguous()
      cfg.n_goal_rows, cfg.goal_table = len(initial_states), self._goal_table.data_ptr()

  def get_next_goal(self):
    """sawyer_peg.py:144-152 (host-side draw; the batched reset draws per env on the device)"""
    if not self._reset_at_goal:
      return self.goal_states[np.random.randint(0, self.goal_states.shape[0])]
    return self.initial_states[np.random.randint(0, self.initial_states.shape[0])].copy()
