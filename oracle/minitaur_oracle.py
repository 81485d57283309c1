"""CPU statement (numpy, fp64, one env at a time) of the minitaur env on this build's articulated-body stepper.

TEST INFRASTRUCTURE, NOT PRODUCT.  **PARITY UNPINNED AND MODEL-LESS**: the reference simulates `pybullet_data/quadruped/minitaur.urdf` in
PyBullet 3.2.0; neither is in /root/reference.  The rigid-body model is this build's own authoring (tools/minitaur_model.py), the
dynamics are this build's stepper (oracle/physics_oracle.LinkModel: MuJoCo-style soft constraints, not Bullet's sequential-impulse solver).
What IS the reference's own code and is followed line by line (and pinned bit-exact by tests/test_glue*.py against goldens recorded from it):

  GoalConditionedMinitaurBulletEnv.step / reset / _reward / is_successful / _get_observation   earl_benchmark/envs/minitaur_gym_env.py:466-546
  MinitaurBulletEnv.step (action check, leg model, action_repeat x (ApplyAction, stepSimulation)), reset (100 settle steps)   :222-270, 276-329
  Minitaur.ApplyAction (velocity-limit clip, motor model, overheat protection, torque with motor direction), GetObservation   envs/minitaur.py:300-390
  Minitaur.ConvertFromLegModel :434-457;  MotorModel.convert_to_torque  envs/motor.py:49-94   (oracle.glue_oracle: leg_to_motor, motor_torque)
  MinitaurEnvRandomizer.randomize_env [UPSTREAM pybullet_envs.bullet.minitaur_env_randomizer, from memory] per reset: base mass x U(0.8, 1.2), leg-link
  and motor mass x U(0.8, 1.2) each, battery voltage U(14.8, 16.8), motor viscous damping U(0, 0.01), foot friction U(0.8, 1.5) -- through the
  reference's own setters Minitaur.SetBaseMass / SetLegMasses / SetFootFriction / SetBatteryVoltage / SetMotorViscousDamping (envs/minitaur.py:468-508).
"""
import os

import numpy as np

from . import glue_oracle as go
from .physics_oracle import LinkModel
from .tabletop_oracle import philox4x32_10

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL = os.path.join(HERE, '..', 'earl_benchmark_amd', 'models', 'minitaur_links.npz')
NUM_SUBSTEPS = 5                                 # minitaur_gym_env.py:25, 161-164 (action_repeat = 1 * NUM_SUBSTEPS)
TIME_STEP = 0.01 / NUM_SUBSTEPS                  # :126, 162
MOTOR_KP, MOTOR_KD = 1.0, 0.02                   # :85-86
MOTOR_VELOCITY_LIMIT = 150.0                     # GoalConditionedMinitaurBulletEnv: :472
DISTANCE_WEIGHT, ENERGY_WEIGHT = 2.0, 0.005      # :473, :71
OVERHEAT_SHUTDOWN_TORQUE, OVERHEAT_SHUTDOWN_TIME = 2.45, 1.0   # minitaur.py:14-15
SETTLE_STEPS = 100                               # minitaur_gym_env.py:265-269
GOAL_LOCATIONS = np.array([[0.4, 0.2], [0.2, 0.2], [-0.2, 0.2], [-0.4, 0.2], [0.4, 0.0], [0.2, 0.0], [-0.2, 0.0], [-0.4, 0.0],
                           [0.4, 0.4], [0.2, 0.4], [-0.2, 0.4], [-0.4, 0.4]])   # :467-469
STREAM_RESET = 0x4D00                            # Philox draw ids of the reset: goal index, battery voltage, viscous damping, base / leg-link / motor mass, foot friction
BASE_MASS_ERR, LEG_MASS_ERR, FOOT_FRICTION = (-0.2, 0.2), (-0.2, 0.2), (0.8, 1.5)   # [UPSTREAM] MINITAUR_BASE_MASS_ERROR_RANGE, ..._LEG_MASS_..., MINITAUR_LEG_FRICTION
ACTION_BOUND, ACTION_EPS = 1.0, 0.01             # :144, 30


def u01(lo, hi):
  return float((((int(hi) << 32) | int(lo)) >> 11) * (1.0 / 9007199254740992.0))


class MinitaurOracle:
  """one env instance (global id `env_id`): state = (qpos [23], qvel [22], overheat counters, enabled flags, motor parameters, goal)"""

  def __init__(self, env_id=0, seed=0, randomize=True, contacts=True):
    self.lm = LinkModel(MODEL)
    self.lm.contacts = contacts
    self.mass0, self.inertia0 = self.lm.mass.copy(), self.lm.inertia.copy()       # the model's own: the randomizer scales from these at every reset
    root = int(self.lm.ball_dof) + 2
    self.base_link = root
    self.upper_links = [l for l in range(root + 1, self.lm.nv) if self.lm.parent[l] == root]          # motor + upper leg link
    self.lower_links = [l for l in range(root + 1, self.lm.nv) if self.lm.parent[l] != root]          # FOOT_LINK_ID (minitaur.py:25)
    self.scale = np.ones(3)
    self.foot_mu = -1.0
    self.env_id, self.seed, self.counter, self.randomize = int(env_id), int(seed), 0, randomize          # True = everything (7), or the bit mask of earl_minitaur_cfg.randomize
    self.motor_dof = [int(x) for x in self.lm.motor_dof]
    self.dir = np.asarray(self.lm.motor_direction, float)
    self.goal = GOAL_LOCATIONS[0].copy()
    self.voltage, self.viscous = 16.0, 0.0
    self.steps = 0
    self.reset()

  # ------------------------------------------------------------------ Minitaur (envs/minitaur.py)
  def motor_angles(self):
    return np.array([self.qpos[self.lm.qadr[d]] for d in self.motor_dof]) * self.dir        # GetMotorAngles :392-404

  def motor_velocities(self):
    return np.array([self.qvel[d] for d in self.motor_dof]) * self.dir                      # GetMotorVelocities :406-418

  def apply_action(self, motor_commands):
    """ApplyAction :326-390 with accurate_motor_model_enabled, motor_overheat_protection, motor_velocity_limit < inf -> generalized forces"""
    q = self.motor_angles()
    cmd = np.clip(np.asarray(motor_commands, float), q - TIME_STEP * MOTOR_VELOCITY_LIMIT, q + TIME_STEP * MOTOR_VELOCITY_LIMIT)   # :339-343
    qdot = self.motor_velocities()
    actual, observed = go.motor_torque(cmd, q, qdot, kp=MOTOR_KP, kd=MOTOR_KD, voltage=self.voltage, viscous_damping=self.viscous)
    for i in range(8):                                                                       # :351-358
      if abs(actual[i]) > OVERHEAT_SHUTDOWN_TORQUE:
        self.overheat[i] += 1
      else:
        self.overheat[i] = 0
      if self.overheat[i] > OVERHEAT_SHUTDOWN_TIME / TIME_STEP:
        self.enabled[i] = False
    self.observed_torques = observed                                                         # :362
    applied = actual * self.dir                                                              # :365
    qfrc = np.zeros(self.lm.nv)
    for i, d in enumerate(self.motor_dof):
      qfrc[d] = applied[i] if self.enabled[i] else 0.0                                       # :367-373
    return qfrc

  def sim_step(self, qfrc, a_prev=None):
    self.qpos, self.qvel, out = self.lm.step(self.qpos, self.qvel, np.zeros(0), np.zeros(3), np.array([1.0, 0, 0, 0]), a_prev=a_prev, qfrc=qfrc)
    return out

  def observation(self):
    """GetObservation :300-324 + the goal (:541-544): motor angles, velocities, torques, base orientation (x, y, z, w), base xy, goal"""
    qw, qx, qy, qz = self.qpos[3:7]
    return np.concatenate([self.motor_angles(), self.motor_velocities(), self.observed_torques, [qx, qy, qz, qw], self.qpos[0:2], self.goal])

  # ------------------------------------------------------------------ env (envs/minitaur_gym_env.py)
  def draw(self, k):
    b = philox4x32_10((STREAM_RESET + k, self.env_id, self.counter & 0xFFFFFFFF, self.counter >> 32), (self.seed & 0xFFFFFFFF, self.seed >> 32))
    return u01(b[0], b[1])

  def reset(self, goal_idx=None):
    gi = min(int(self.draw(0) * len(GOAL_LOCATIONS)), len(GOAL_LOCATIONS) - 1) if goal_idx is None else int(goal_idx)    # get_next_goal :490-493
    self.goal = GOAL_LOCATIONS[gi].copy()
    self.goal_idx = gi
    rnd = 7 if self.randomize is True else (int(self.randomize) if self.randomize else 0)
    if rnd & 1:                                          # SetBatteryVoltage, SetMotorViscousDamping (minitaur.py:500-508)
      self.voltage = 14.8 + (16.8 - 14.8) * self.draw(1)
      self.viscous = 0.01 * self.draw(2)
    if rnd & 2:                                          # SetBaseMass, SetLegMasses (:468-488): the first leg mass goes to ALL 16 leg links, upper and lower
      lm = self.lm
      leg0, motor0 = float(lm.rand_leg_mass), float(lm.rand_motor_mass)
      s_base = 1.0 + BASE_MASS_ERR[0] + (BASE_MASS_ERR[1] - BASE_MASS_ERR[0]) * self.draw(3)
      leg = leg0 * (1.0 + LEG_MASS_ERR[0] + (LEG_MASS_ERR[1] - LEG_MASS_ERR[0]) * self.draw(4))
      motor = motor0 * (1.0 + LEG_MASS_ERR[0] + (LEG_MASS_ERR[1] - LEG_MASS_ERR[0]) * self.draw(5))
      self.scale = np.array([s_base, (motor + leg) / self.mass0[self.upper_links[0]], leg / self.mass0[self.lower_links[0]]])
      for links, f in (([self.base_link], self.scale[0]), (self.upper_links, self.scale[1]), (self.lower_links, self.scale[2])):
        for l in links:
          lm.mass[l], lm.inertia[l] = self.mass0[l] * f, self.inertia0[l] * f
    if rnd & 4:                                          # SetFootFriction (:490-498): every contact of a lower-leg link
      self.foot_mu = FOOT_FRICTION[0] + (FOOT_FRICTION[1] - FOOT_FRICTION[0]) * self.draw(6)
      self.lm.link_mu = {l: self.foot_mu for l in self.lower_links}
    self.counter += 1
    self.qpos, self.qvel = np.array(self.lm.qpos0, float), np.zeros(self.lm.nv)             # Minitaur.Reset(reload_urdf=False) :170-176
    self.overheat, self.enabled = np.zeros(8, int), [True] * 8                               # :178-179
    self.observed_torques = np.zeros(8)
    self.steps = 0
    a_prev = None
    for _ in range(SETTLE_STEPS):                                                            # minitaur_gym_env.py:265-269
      qfrc = self.apply_action([np.pi / 2] * 8)
      a_prev = self.sim_step(qfrc, a_prev)['qacc']
    return self.observation()

  def step(self, action):
    a = np.asarray(action, float)
    if ((a < -ACTION_BOUND - ACTION_EPS) | (a > ACTION_BOUND + ACTION_EPS)).any():           # :276-281
      raise ValueError('action out of bounds')
    cmd = go.leg_to_motor(a[None])[0]                                                        # ConvertFromLegModel
    a_prev = None                                                                            # (every env step starts the active-set iteration cold)
    for _ in range(NUM_SUBSTEPS):                                                            # :321-323
      qfrc = self.apply_action(cmd)
      a_prev = self.sim_step(qfrc, a_prev)['qacc']
    self.steps += 1
    obs = self.observation()
    x_dist, y_dist = self.qpos[0] - self.goal[0], self.qpos[1] - self.goal[1]               # _reward :505-521
    energy = abs(float(np.dot(self.observed_torques, self.motor_velocities()))) * TIME_STEP
    reward = DISTANCE_WEIGHT * (-abs(x_dist) - abs(y_dist)) - ENERGY_WEIGHT * energy
    success = float(np.sqrt(np.sum((obs[28:30] - obs[30:32]) ** 2)) < 0.1)                   # is_successful :495-503
    return obs, reward, False, {'success': success}
