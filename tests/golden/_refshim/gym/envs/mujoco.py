import xml.etree.ElementTree as ET

import numpy as np


class _Data:
  def __init__(self, nq, nv):
    self.qpos = np.zeros(nq)
    self.qvel = np.zeros(nv)


class _Sim:
  def __init__(self, nq, nv):
    self.data = _Data(nq, nv)

  def forward(self):  # mj_forward never writes qpos/qvel
    pass


class MujocoEnv:
  """Holds qpos/qvel; nq is counted from the <worldbody> joints of the model file."""

  def __init__(self, model_path, frame_skip):
    # all tabletop joints are 1-dof slides, so nq = nv = number of <joint> under <worldbody>
    world = ET.parse(model_path).getroot().find('worldbody')
    nq = sum(1 for _ in world.iter('joint'))
    self.frame_skip = frame_skip
    self.sim = _Sim(nq, nq)

  def set_state(self, qpos, qvel):
    assert qpos.shape == self.sim.data.qpos.shape, (qpos.shape, self.sim.data.qpos.shape)
    self.sim.data.qpos = np.array(qpos, dtype=np.float64)
    self.sim.data.qvel = np.array(qvel, dtype=np.float64)
