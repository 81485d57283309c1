"""CPU restatement of the Sawyer door env glue around oracle/physics_oracle.LinkModel.  TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED for the dynamics (see physics_oracle.py).  The env glue restated here:
  SawyerXYZEnv.step / set_xyz_action / _reset_hand   [UPSTREAM metaworld, not in /root/reference; SURVEY.md Appendix D]
  SawyerDoorV2._get_obs          /root/reference/earl_benchmark/envs/sawyer_door.py:86-94
  SawyerDoorV2.reset_model       :111-125
  SawyerDoorV2.compute_reward    :141-171, is_successful :173-177
  PersistentStateWrapper.step    /root/reference/earl_benchmark/wrappers/persistent_state_wrapper.py:17-31
Pinned by the reference's data: the sparse rule on all 1095 demonstration rows, goal / initial-state constants.
"""
import numpy as np

from .tabletop_oracle import philox4x32_10  # same draw layout as the kernels: ctr = (draw, env id, counter lo, hi), key = seed

GOAL = np.array([0.29072163, 0.74286009, 0.10003595, 1.0, 0.29072163, 0.74286009, 0.10003595])   # sawyer_door.py:15-16
MOCAP_LOW, MOCAP_HIGH = np.array([-0.5, 0.40, 0.05]), np.array([0.5, 1.0, 0.5])                    # :25-26
MOCAP_QUAT = np.array([1.0, 0.0, 1.0, 0.0])
# timesteps of the reset settle: the converged pose, not the reference's 250-timestep transient (see earl_benchmark_amd/envs/sawyer_door.py:
# the demonstrations show MuJoCo ending that transient with a vertical gripper; this stepper's transient takes 1,000 timesteps to get there)
SETTLE_TIMESTEPS = 2000


def recorded_reset(lm, hand_init_pos, default_hand, q, v, reset_state):
  """Round 4: the arm's seven angles and speeds at the start of every recorded episode of the task (tables reset_qpos_recorded / reset_qvel_recorded,
  tools/weld_free_motion_fit.py) replace the converged ones when the env resets the hand where the recordings do; fingers and object keep the settled values.
  Mirrors earl_benchmark_amd/envs/sawyer_door.py _settle_reset_hand."""
  if reset_state == 'recorded' and hasattr(lm, 'reset_qpos_recorded') and np.allclose(hand_init_pos, default_hand):
    q, v = q.copy(), v.copy()
    q[:7], v[:7] = lm.reset_qpos_recorded, lm.reset_qvel_recorded
  return q, v
OBJ_INIT_POS = np.array([0.1, 0.95, 0.1], dtype=np.float32).astype(np.float64)                     # :36
BAD_VALUE = 1e10            # EARL_BAD_VALUE (MuJoCo's mjMAXVAL): beyond it, or NaN, an env step is rolled back


def u01(lo, hi):
  return float(((int(hi) << 32) | int(lo)) >> 11) * (1.0 / 9007199254740992.0)


def tolerance_gaussian(x, hi, margin):
  """metaworld reward_utils.tolerance(x, bounds=(0, hi), margin, sigmoid='gaussian', value_at_margin=0.1) [UPSTREAM]"""
  if 0.0 <= x <= hi:
    return 1.0
  if margin == 0:
    return 0.0
  d = (-x if x < 0 else x - hi) / margin
  scale = np.sqrt(-2.0 * np.log(0.1))
  return float(np.exp(-0.5 * (d * scale) ** 2))


def compute_reward(obs, reward_type, hand_init_pos):
  tcp, obj, target = obs[:3], obs[4:7], obs[11:14]
  obj_to_target = float(np.sqrt(np.sum((obj - target) ** 2)))
  ok = obj_to_target <= 0.02
  if reward_type == 'sparse':
    return float(ok), ok
  tcp_to_obj = float(np.sqrt(np.sum((tcp - obj) ** 2)))
  in_place = tolerance_gaussian(obj_to_target, 0.05, float(np.sqrt(np.sum((OBJ_INIT_POS - target) ** 2))))
  hand_in_place = tolerance_gaussian(tcp_to_obj, 0.25 * 0.05, float(np.sqrt(np.sum((hand_init_pos - obj) ** 2))) + 0.1)
  r = 3 * hand_in_place + 6 * in_place
  if obj_to_target < 0.05:
    r = 10.0
  return float(r), ok


def door_info(obs, reward_type, hand_init_pos):
  """the info dict of SawyerDoorV2.step = evaluate_state (reference sawyer_door.py:127-139): compute_reward returns [reward, obj_to_target, hand_in_place]
  (:171), so 'in_place_reward' carries the HAND's tolerance term; 'success' is obj_to_target <= 0.08, not is_successful()'s 0.02"""
  tcp, obj, target = obs[:3], obs[4:7], obs[11:14]
  obj_to_target = float(np.sqrt(np.sum((obj - target) ** 2)))
  tcp_to_obj = float(np.sqrt(np.sum((tcp - obj) ** 2)))
  hand_in_place = tolerance_gaussian(tcp_to_obj, 0.25 * 0.05, float(np.sqrt(np.sum((hand_init_pos - obj) ** 2))) + 0.1)
  reward, _ = compute_reward(obs, reward_type, hand_init_pos)
  return {'obj_to_target': obj_to_target, 'in_place_reward': hand_in_place, 'success': float(obj_to_target <= 0.08), 'near_object': 0.0,
          'grasp_success': 1.0, 'grasp_reward': 1.0, 'unscaled_reward': reward}


class SawyerDoorOracle:
  """one env instance (global id `env_id`), scalar loops"""

  def __init__(self, link_model, reward_type='sparse', reset_at_goal=False, seed=0, env_id=0, horizon=0, frame_skip=5, reset_state='recorded'):
    self.lm, self.reward_type, self.seed, self.env_id, self.horizon, self.frame_skip = link_model, reward_type, seed, env_id, horizon, frame_skip
    self.reset_state = reset_state
    names = [str(x) for x in link_model.att_names]
    self.k_hand, self.k_right, self.k_left, self.k_obj = (names.index(x) for x in ('hand', 'rightEndEffector', 'leftEndEffector', 'handle'))
    self.obj_dof = link_model.nv - 1
    self.hand_init_pos = np.array([0.29, 0.74, 0.1] if reset_at_goal else [0, 0.4, 0.2], dtype=np.float32).astype(np.float64)
    self.obj_init_angle = 0.0 if reset_at_goal else -np.pi / 3
    self.angle_noise = (-np.pi / 20, 0.0) if reset_at_goal else (0.0, np.pi / 20)
    self.goal = GOAL.copy()
    self.counter = 0
    self._settled = None

  def settle(self):
    if self._settled is None:
      q, v = np.zeros(self.lm.nv), np.zeros(self.lm.nv)
      for _ in range(SETTLE_TIMESTEPS):
        q, v, _ = self.lm.step(q, v, np.array([-1.0, 1.0]), self.hand_init_pos, MOCAP_QUAT)
      self._settled = recorded_reset(self.lm, self.hand_init_pos, [0, 0.4, 0.2], q, v, self.reset_state)
    return self._settled

  def obs_from(self, pos, quat):
    at = lambda k: self.lm.attachment(pos, quat, k)[0]
    grip = float(np.clip(np.sqrt(np.sum((at(self.k_right) - at(self.k_left)) ** 2)) / 0.1, 0.0, 1.0))
    return np.concatenate([at(self.k_hand), [grip], at(self.k_obj), self.goal])

  def reset(self):
    q, v = (x.copy() for x in self.settle())
    b = philox4x32_10((0, self.env_id, self.counter & 0xFFFFFFFF, self.counter >> 32), (self.seed & 0xFFFFFFFF, self.seed >> 32))
    self.counter += 1
    q[self.obj_dof] = self.obj_init_angle + (self.angle_noise[0] + (self.angle_noise[1] - self.angle_noise[0]) * u01(b[0], b[1]))
    v[self.obj_dof] = 0.0
    self.qpos, self.qvel, self.mocap, self.steps = q, v, self.hand_init_pos.copy(), 0
    pos, quat, _ = self.lm.kinematics(q)
    self.last_obs = self.obs_from(pos, quat)
    return self.last_obs.copy()

  def step(self, action):
    a = np.asarray(action, dtype=np.float32)
    delta = (np.clip(a[:3], np.float32(-1), np.float32(1)) * np.float32(1.0 / 100)).astype(np.float64)   # f32 product, as numpy computes it
    mocap0 = self.mocap
    self.mocap = np.clip(self.mocap + delta, MOCAP_LOW, MOCAP_HIGH)
    ctrl = np.array([float(a[3]), -float(a[3])])
    out = None
    stable = (self.qpos.copy(), self.qvel.copy(), mocap0)
    for _ in range(self.frame_skip):                       # (every env step starts cold, later timesteps from the previous solution)
      self.qpos, self.qvel, out = self.lm.step(self.qpos, self.qvel, ctrl, self.mocap, MOCAP_QUAT, None if out is None else out['qacc'])
    self.steps += 1
    done = bool(self.horizon > 0 and self.steps >= self.horizon)
    if self._diverged():                                    # failure guard: include/earl_physics.h, earl_sawyer_out.status
      self.qpos, self.qvel, self.mocap = stable
      self.fail_count += 1
      return self.last_obs.copy(), np.float32(0.0), done, False
    obs = self.obs_from(out['pos'], out['quat'])            # mj_step leaves the kinematics of the last timestep's START
    r, ok = compute_reward(obs, self.reward_type, self.hand_init_pos)
    self.last_obs = obs.copy()
    return obs, np.float32(r), done, bool(ok)

  fail_count = 0

  def _diverged(self):
    return not (np.all(np.abs(self.qpos) < BAD_VALUE) and np.all(np.abs(self.qvel) < BAD_VALUE))


# ---------------------------------------------------------------------------------------------------- sawyer_peg
PEG_GOAL = np.array([0.0, 0.6, 0.2, 1.0, -0.3 + 0.03, 0.6, 0.0 + 0.13])                            # sawyer_peg.py:52
PEG_INITIAL_XY = np.array([(0.00313463, 0.68326396), (-0.04035005, 0.67949003), (0.02531051, 0.6074387), (0.05957219, 0.6271171),
                           (-0.07566337, 0.62575287), (-0.01177235, 0.55206996), (0.02779735, 0.54707706), (0.01835314, 0.5329686),
                           (0.02690855, 0.6263067), (0.01766127, 0.59630984), (0.0560186, 0.6634998), (-0.03950658, 0.6323736),
                           (-0.03216827, 0.5247563), (0.01265727, 0.69466716), (0.05076993, 0.6025737)])            # :18-48
PEG_WIDE_INITIAL = np.array([[-0.3, 0.8, 0.02], [-0.4, 0.8, 0.02], [-0.3, 0.9, 0.02], [-0.4, 0.9, 0.02], [-0.2, 0.8, 0.02], [-0.2, 0.75, 0.02],
                             [-0.2, 0.9, 0.02], [-0.1, 0.77, 0.02], [0.0, 0.9, 0.02], [0.1, 0.8, 0.02], [0.15, 0.75, 0.02], [-0.3, 0.4, 0.02],
                             [-0.4, 0.4, 0.02], [-0.3, 0.45, 0.02], [-0.4, 0.45, 0.02], [-0.2, 0.4, 0.02], [-0.2, 0.45, 0.02], [-0.2, 0.38, 0.02],
                             [-0.1, 0.42, 0.02], [0.0, 0.45, 0.02], [0.1, 0.36, 0.02], [0.15, 0.44, 0.02]])                    # :55-60
PEG_INITIAL_STATES = np.array([[0.00615235, 0.6001898, 0.19430117, 1.0, x, y, 0.02] for x, y in PEG_INITIAL_XY])


# ---- metaworld reward_utils / SawyerXYZEnv._gripper_caging_reward [UPSTREAM metaworld master, NOT in /root/reference: restated from
# the published source as this builder knows it; UNPINNED].  The reference calls them at sawyer_peg.py:246-249 (tolerance, long_tail),
# :256-265 (rect_prism_tolerance, hamacher_product), :276-282 (_gripper_caging_reward), :284-286.
def tolerance_long_tail(x, lo, hi, margin):
  """reward_utils.tolerance(x, bounds=(lo, hi), margin, sigmoid='long_tail', value_at_margin=0.1)"""
  if lo <= x <= hi:
    return 1.0
  if margin == 0:
    return 0.0
  d = (lo - x if x < lo else x - hi) / margin
  scale = np.sqrt(1 / 0.1 - 1)
  return float(1 / ((d * scale) ** 2 + 1))


def rect_prism_tolerance(curr, zero, one):
  in_range = lambda a, b, c: (b <= a <= c) if c >= b else (c <= a <= b)
  if in_range(curr[0], zero[0], one[0]) and in_range(curr[1], zero[1], one[1]) and in_range(curr[2], zero[2], one[2]):
    diff = one - zero
    return float((curr[0] - zero[0]) / diff[0] * ((curr[1] - zero[1]) / diff[1]) * ((curr[2] - zero[2]) / diff[2]))
  return 1.0


def hamacher_product(a, b):
  den = a + b - (a * b)
  return float((a * b) / den) if den > 0 else 0.0


def gripper_caging_reward(action, obj_pos, left_pad, right_pad, tcp, obj_init_pos, init_tcp, obj_radius, pad_success_thresh, xz_thresh):
  """SawyerXYZEnv._gripper_caging_reward(..., high_density=True) [UPSTREAM]"""
  pad_y_lr = np.array([left_pad[1], right_pad[1]])
  pad_to_obj_lr = np.abs(pad_y_lr - obj_pos[1])
  pad_to_objinit_lr = np.abs(pad_y_lr - obj_init_pos[1])
  caging_lr_margin = np.abs(pad_to_objinit_lr - pad_success_thresh)
  caging_lr = [tolerance_long_tail(pad_to_obj_lr[i], obj_radius, pad_success_thresh, caging_lr_margin[i]) for i in range(2)]
  caging_y = hamacher_product(*caging_lr)
  xz = [0, 2]
  caging_xz_margin = np.sqrt(np.sum((obj_init_pos[xz] - init_tcp[xz]) ** 2)) - xz_thresh
  caging_xz = tolerance_long_tail(np.sqrt(np.sum((tcp[xz] - obj_pos[xz]) ** 2)), 0.0, xz_thresh, caging_xz_margin)
  gripper_closed = min(max(0.0, float(action[-1])), 1.0) / 1.0
  caging = hamacher_product(caging_y, caging_xz)
  gripping = gripper_closed if caging > 0.97 else 0.0
  caging_and_gripping = hamacher_product(caging, gripping)
  return (caging_and_gripping + caging) / 2          # high_density


def peg_info(obs, action, reward_type, grasp_site, head_site, left_pad, right_pad, tcp_center, obj_init_pos, peg_head_pos_init, init_tcp, corners):
  """the info dict of SawyerPegV2.step = evaluate_state (reference sawyer_peg.py:165-184) around compute_reward (:231-299), either reward type"""
  terms = {}
  dense = peg_dense_reward(obs, action, grasp_site, head_site, left_pad, right_pad, tcp_center, obj_init_pos, peg_head_pos_init, init_tcp, corners,
                           dense=reward_type == 'dense', terms=terms)
  ok = bool(np.sqrt(np.sum((obs[4:7] - obs[11:14]) ** 2)) <= 0.05)                      # is_successful :301-305
  reward = dense if reward_type == 'dense' else float(ok)                             # :296-297
  return {'success': float(terms['obj_to_target'] <= 0.05), 'near_object': float(terms['tcp_to_obj'] <= 0.03),
          'grasp_success': float(terms['tcp_to_obj'] < 0.02 and obs[3] > 0 and obs[6] - 0.01 > obj_init_pos[2]),
          'grasp_reward': terms['object_grasped'], 'in_place_reward': terms['in_place'], 'obj_to_target': terms['obj_to_target'], 'unscaled_reward': reward}


def peg_dense_reward(obs, action, grasp_site, head_site, left_pad, right_pad, tcp_center, obj_init_pos, peg_head_pos_init, init_tcp, corners, dense=True,
                     terms=None):
  """SawyerPegV2.compute_reward with reward_type='dense' (sawyer_peg.py:231-299); obs[4:7] is the pegHead site, so
  obj = obs[4:7] - pegHead + pegGrasp (:233-234).  dense=False: the 'sparse' branch of :284-285 (object_grasped = 0 unless lifted) -- the value the
  function then returns is NOT the sparse reward (that is float(is_successful), :296-297), only the terms are of interest."""
  tcp, obj_head, tcp_opened, target = obs[:3], obs[4:7], obs[3], obs[11:14]
  obj = obs[4:7] - head_site + grasp_site
  tcp_to_obj = np.sqrt(np.sum((obj - tcp) ** 2))
  scale = np.array([1.0, 2.0, 2.0])
  obj_to_target = np.sqrt(np.sum(((obj_head - target) * scale) ** 2))
  in_place_margin = np.sqrt(np.sum(((peg_head_pos_init - target) * scale) ** 2))
  in_place = tolerance_long_tail(obj_to_target, 0.0, 0.05, in_place_margin)
  box1 = rect_prism_tolerance(obj_head, corners[0], corners[1])      # curr, zero = bottom-right corner, one = top-left corner (:258-263)
  box2 = rect_prism_tolerance(obj_head, corners[2], corners[3])
  in_place = hamacher_product(in_place, hamacher_product(box2, box1))
  lifted = tcp_to_obj < 0.08 and tcp_opened > 0 and obj[2] - 0.01 > obj_init_pos[2]
  if lifted:
    object_grasped = 1.0
  elif dense:
    # the caging reward reads self.tcp_center (midpoint of the two finger sites), not the hand position of the observation
    object_grasped = gripper_caging_reward(action, obj, left_pad, right_pad, tcp_center, obj_init_pos, init_tcp, 0.0075, 0.03, 0.005)
  else:
    object_grasped = 0.0
  reward = hamacher_product(object_grasped, in_place)
  if lifted:
    reward += 1.0 + 5 * in_place
  if obj_to_target <= 0.05:
    reward = 10.0
  if terms is not None:
    terms.update(tcp_to_obj=float(tcp_to_obj), obj_to_target=float(obj_to_target), object_grasped=float(object_grasped), in_place=float(in_place))
  return float(reward)


class SawyerPegOracle(SawyerDoorOracle):
  """SawyerPegV2 (sparse reward) around LinkModel: reference earl_benchmark/envs/sawyer_peg.py
     _get_obs :134-142 (object = site pegHead :186-187), reset_model :192-229, get_next_goal / reset_goal :144-163,
     is_successful :301-305 (radius 0.05); SawyerXYZEnv.step / _reset_hand / _set_obj_xyz are upstream metaworld."""

  def __init__(self, link_model, reward_type='sparse', reset_at_goal=False, seed=0, env_id=0, horizon=0, frame_skip=5, wide_init=False,
               goal_change_frequency=0, reset_state='recorded'):
    self.wide_init, self.gcf, self.sgc, self.total_steps = wide_init, goal_change_frequency, 0, 0
    self.reset_state = reset_state
    self.lm, self.reward_type, self.seed, self.env_id, self.horizon, self.frame_skip = link_model, reward_type, seed, env_id, horizon, frame_skip
    names = [str(x) for x in link_model.att_names]
    self.k_hand, self.k_right, self.k_left, self.k_obj = (names.index(x) for x in ('hand', 'rightEndEffector', 'leftEndEffector', 'pegHead'))
    self.obj_dof = int(link_model.ball_dof) - 3
    self.hand_init_pos = np.array([0, 0.6, 0.2], dtype=np.float64)
    self.reset_at_goal = reset_at_goal
    pos_box = PEG_GOAL[4:] - np.array([0.03, 0.0, 0.13])
    if not reset_at_goal:
      self.obj_low, self.obj_high = np.array([0.0, 0.5, 0.02]), np.array([0.2, 0.7, 0.02])
      self.reject_xy, self.reject_radius = pos_box[:2], 0.1
    else:
      goal_pos = PEG_GOAL[4:] - np.array([-0.1, 0.0, 0.0])
      self.obj_low, self.obj_high = goal_pos - 0.02, goal_pos + 0.02
      self.reject_xy, self.reject_radius = pos_box[:2], 0.0
    self.goal = PEG_GOAL.copy()
    self.counter = 0
    self._settled = None

  def settle(self):
    if self._settled is None:
      q, v = self.lm.qpos0.copy(), np.zeros(self.lm.nv)
      for _ in range(SETTLE_TIMESTEPS):
        q, v, _ = self.lm.step(q, v, np.array([-1.0, 1.0]), self.hand_init_pos, MOCAP_QUAT)
      self._settled = recorded_reset(self.lm, self.hand_init_pos, [0, 0.6, 0.2], q, v, self.reset_state)
    return self._settled

  def _draw(self, d):
    return philox4x32_10((d, self.env_id, self.counter & 0xFFFFFFFF, self.counter >> 32), (self.seed & 0xFFFFFFFF, self.seed >> 32))

  def reset(self):
    q, v = (x.copy() for x in self.settle())
    wide = False
    if self.wide_init and not self.reset_at_goal:       # sawyer_peg.py:200-209
      c0, c1, c2 = self._draw(0xFFF0), self._draw(0xFFF1), self._draw(0xFFF2)
      wide = not (u01(c0[0], c0[1]) < 0.5)
      row = PEG_WIDE_INITIAL[min(int(u01(c0[2], c0[3]) * len(PEG_WIDE_INITIAL)), len(PEG_WIDE_INITIAL) - 1)]
      un = np.array([u01(c1[0], c1[1]), u01(c1[2], c1[3]), u01(c2[0], c2[1])])
      pos = (row + np.array([0.1, 0.0, 0.0])) + (-0.02 + (0.02 - -0.02) * un)
    for attempt in range(0 if wide else 16):
      b0, b1 = self._draw(2 * attempt), self._draw(2 * attempt + 1)
      u = np.array([u01(b0[0], b0[1]), u01(b0[2], b0[3]), u01(b1[0], b1[1])])
      pos = self.obj_low + (self.obj_high - self.obj_low) * u
      if not np.sqrt(np.sum((pos[:2] - self.reject_xy) ** 2)) < self.reject_radius:
        break
    if self.reset_at_goal:
      b = self._draw(0xFFFF)
      self.goal = PEG_INITIAL_STATES[min(int(u01(b[0], b[1]) * len(PEG_INITIAL_STATES)), len(PEG_INITIAL_STATES) - 1)].copy()
    self.counter += 1
    q[self.obj_dof:self.obj_dof + 3] = pos          # _set_obj_xyz [UPSTREAM]: qpos[9:12] <- pos, qvel[9:15] <- 0
    v[self.obj_dof:self.obj_dof + 6] = 0.0
    self.qpos, self.qvel, self.mocap, self.steps, self.sgc = q, v, self.hand_init_pos.copy(), 0, 0
    pos_, quat_, _ = self.lm.kinematics(q)
    # state the dense reward keeps from the reset (sawyer_peg.py:214-215; init_tcp: SawyerXYZEnv._reset_hand [UPSTREAM])
    names = [str(x) for x in self.lm.att_names]
    at = lambda n, P=None, Q=None: self.lm.attachment(pos_ if P is None else P, quat_ if Q is None else Q, names.index(n))[0]
    self.obj_init_pos = pos.copy()
    self.peg_head_pos_init = at('pegHead')
    ps, qs, _ = self.lm.kinematics(self.settle()[0])
    self.init_tcp = 0.5 * (at('rightEndEffector', ps, qs) + at('leftEndEffector', ps, qs))
    self.last_obs = self.obs_from(pos_, quat_)
    return self.last_obs.copy()

  def step(self, action):
    a = np.asarray(action, dtype=np.float32)
    delta = (np.clip(a[:3], np.float32(-1), np.float32(1)) * np.float32(1.0 / 100)).astype(np.float64)
    stable = (self.qpos.copy(), self.qvel.copy(), self.mocap)
    self.mocap = np.clip(self.mocap + delta, MOCAP_LOW, MOCAP_HIGH)
    ctrl = np.array([float(a[3]), -float(a[3])])
    out = None
    for _ in range(self.frame_skip):                       # (every env step starts cold, later timesteps from the previous solution)
      self.qpos, self.qvel, out = self.lm.step(self.qpos, self.qvel, ctrl, self.mocap, MOCAP_QUAT, None if out is None else out['qacc'])
    failed = self._diverged()                            # failure guard: include/earl_physics.h, earl_sawyer_out.status
    if failed:
      self.qpos, self.qvel, self.mocap = stable
      self.fail_count += 1
      obs = self.last_obs.copy()
    else:
      obs = self.obs_from(out['pos'], out['quat'])
      self.last_obs = obs.copy()
    ok = (not failed) and bool(np.sqrt(np.sum((obs[4:7] - obs[11:14]) ** 2)) <= 0.05)             # is_successful :301-305
    rew = float(ok)
    self.last_info = None
    if not failed:
      names = [str(x) for x in self.lm.att_names]
      at = lambda n: self.lm.attachment(out['pos'], out['quat'], names.index(n))[0]
      sites = (at('pegGrasp'), at('pegHead'), at('leftpad'), at('rightpad'), 0.5 * (at('rightEndEffector') + at('leftEndEffector')))
      if self.reward_type == 'dense':
        rew = peg_dense_reward(obs, a, *sites, self.obj_init_pos, self.peg_head_pos_init, self.init_tcp, self.lm.peg_box_corners)
      self.last_info = peg_info(obs, a, self.reward_type, *sites, self.obj_init_pos, self.peg_head_pos_init, self.init_tcp, self.lm.peg_box_corners)
    self.steps += 1
    if self.gcf > 0:                                    # LifelongWrapper.step (lifelong_wrapper.py:30-44)
      self.sgc += 1
      if self.sgc >= self.gcf:
        self.sgc = 0
        if self.reset_at_goal:
          b = philox4x32_10((0xFFFE, self.env_id, self.total_steps & 0xFFFFFFFF, self.total_steps >> 32), (self.seed & 0xFFFFFFFF, self.seed >> 32))
          self.goal = PEG_INITIAL_STATES[min(int(u01(b[0], b[1]) * len(PEG_INITIAL_STATES)), len(PEG_INITIAL_STATES) - 1)].copy()
          obs = np.concatenate([obs[:7], self.goal])
    self.total_steps += 1
    return obs, np.float32(rew), bool(self.horizon > 0 and self.steps >= self.horizon), ok
