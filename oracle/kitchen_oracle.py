"""CPU restatement of the kitchen env step around oracle/physics_oracle.LinkModel (numpy).  TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY WITH MUJOCO UNPINNED (no recording of this env exists in the reference; the simulator cannot run here).  The env glue restated here is
pinned piecewise by goldens recorded from the reference's own methods (tests/test_glue.py):
  KitchenV0.step              /root/reference/earl_benchmark/envs/kitchen_assets/adept_envs/adept_envs/franka/kitchen_multitask_v0.py:91-125
  Robot.step / get_obs        .../franka/robot/franka_robot.py:137-207, :259-264; MujocoEnv.do_simulation .../adept_envs/mujoco_env.py:148-157
  Kitchen._get_reward_n_score /root/reference/earl_benchmark/envs/kitchen.py:141-183
"""
import os

import numpy as np

from . import glue_oracle as go
from . import physics_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAME_SKIP = 40
SITES = ('knob1_site', 'knob2_site', 'knob3_site', 'knob4_site', 'light_site', 'slide_site', 'hinge_site2', 'microhandle_site')


class KitchenOracle:
  """one env instance; the caller supplies the sensor-noise draws (or None), exactly the inputs the HIP glue takes"""

  def __init__(self, params, link_model=None):
    self.lm = link_model or po.LinkModel(os.path.join(ROOT, 'earl_benchmark_amd', 'models', 'kitchen_links.npz'))
    self.p = params
    names = [str(x) for x in self.lm.att_names]
    self.site_idx = [names.index(s) for s in SITES]
    self.mq = np.array(self.lm.weld_mocap_quat)

  def set(self, qpos, qvel, mocap_pos, goal, last_qp_robot):
    self.qpos, self.qvel, self.mocap, self.goal, self.last = (np.array(x, np.float64) for x in (qpos, qvel, mocap_pos, goal, last_qp_robot))

  def step(self, action, noise=None):
    a = np.asarray(action, np.float32).astype(np.float64)[None]
    mp, ctrl9 = go.kitchen_action(self.p, a, self.mocap[None], self.last[None])
    self.mocap = mp[0]
    out = None
    for _ in range(FRAME_SKIP):
      self.qpos, self.qvel, out = self.lm.step(self.qpos, self.qvel, ctrl9[0, :2], self.mocap, self.mq, None if out is None else out['qacc'])
    obs = go.kitchen_obs(self.p, self.qpos[None], self.goal[None], None if noise is None else np.asarray(noise)[None])[0]
    self.last = obs[:9].copy()
    sites = np.stack([self.lm.attachment(out['pos'], out['quat'], k)[0] for k in self.site_idx])     # kinematics of the last timestep's start
    r, s = go.kitchen_reward(obs[None], self.mocap[None], sites[None])
    return obs, float(r[0]), bool(s[0]), out
