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
OBJ_INIT_POS = np.array([0.1, 0.95, 0.1], dtype=np.float32).astype(np.float64)                     # :36


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


class SawyerDoorOracle:
  """one env instance (global id `env_id`), scalar loops"""

  def __init__(self, link_model, reward_type='sparse', reset_at_goal=False, seed=0, env_id=0, horizon=0, frame_skip=5):
    self.lm, self.reward_type, self.seed, self.env_id, self.horizon, self.frame_skip = link_model, reward_type, seed, env_id, horizon, frame_skip
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
      for _ in range(50 * self.frame_skip):
        q, v, _ = self.lm.step(q, v, np.array([-1.0, 1.0]), self.hand_init_pos, MOCAP_QUAT)
      self._settled = (q, v)
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
    return self.obs_from(pos, quat)

  def step(self, action):
    a = np.asarray(action, dtype=np.float32)
    delta = (np.clip(a[:3], np.float32(-1), np.float32(1)) * np.float32(1.0 / 100)).astype(np.float64)   # f32 product, as numpy computes it
    self.mocap = np.clip(self.mocap + delta, MOCAP_LOW, MOCAP_HIGH)
    ctrl = np.array([float(a[3]), -float(a[3])])
    out = None
    for _ in range(self.frame_skip):
      self.qpos, self.qvel, out = self.lm.step(self.qpos, self.qvel, ctrl, self.mocap, MOCAP_QUAT)
    obs = self.obs_from(out['pos'], out['quat'])            # mj_step leaves the kinematics of the last timestep's START
    r, ok = compute_reward(obs, self.reward_type, self.hand_init_pos)
    self.steps += 1
    return obs, np.float32(r), bool(self.horizon > 0 and self.steps >= self.horizon), bool(ok)
