"""HIP articulated-body stepper (csrc/physics.hip) vs its CPU reference (oracle/physics_oracle.py: LinkModel).
Tolerance 1e-9 relative: same algorithm in fp64; device sincos / pow / sqrt and summation order differ in the last bits.
Parity with MuJoCo itself is unpinned."""
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
LINKS = os.path.join(REPO, 'earl_benchmark_amd', 'models', 'sawyer_door_links.npz')


@pytest.fixture(scope='module')
def env():
  import torch
  from earl_benchmark_amd import physics
  from oracle import physics_oracle as po
  return torch, physics.DeviceModel('sawyer_door'), po.LinkModel(LINKS)


def random_states(n, seed):
  rng = np.random.default_rng(seed)
  qpos = rng.uniform(-1, 1, size=(n, 10)) * 0.6
  qpos[:, 1] = rng.uniform(-3.0, -0.3, n)        # straddles the upper limit -0.5
  qpos[:, 7] = rng.uniform(-0.005, 0.045, n); qpos[:, 8] = rng.uniform(-0.035, 0.005, n)   # claws around their stops
  qpos[:, 9] = rng.uniform(-1.5, 0.1, n)
  qvel = rng.normal(size=(n, 10)) * 0.3
  mp = rng.uniform([-0.3, 0.4, 0.05], [0.3, 0.9, 0.45], size=(n, 3))
  mq = np.tile([1.0, 0, 1, 0], (n, 1)) + rng.normal(size=(n, 4)) * 0.05
  ctrl = rng.uniform(-1.3, 1.3, size=(n, 2))
  return qpos, qvel, mp, mq, ctrl


def test_layout(env):
  import ctypes as C
  torch, dm, lm = env
  from earl_benchmark_amd.physics import LinkModelStruct
  assert dm.lib.earl_physics_model_size() == C.sizeof(LinkModelStruct)
  assert dm.att_names == ['hand', 'rightEndEffector', 'leftEndEffector', 'endEffector', 'handle']


def test_forward_matches_reference(env):
  torch, dm, lm = env
  n = 96
  qpos, qvel, mp, mq, ctrl = random_states(n, 0)
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
  qacc, efc, att = dm.forward(t(qpos), t(qvel), t(mp), t(mq), t(ctrl))
  qacc, efc, att = qacc.cpu().numpy(), efc.cpu().numpy(), att.cpu().numpy()
  nlim = 0
  for i in range(n):
    ref = lm.forward(qpos[i], qvel[i], ctrl[i], mp[i], mq[i])
    np.testing.assert_allclose(qacc[i], ref['qacc'], rtol=1e-8, atol=1e-8 * np.abs(ref['qacc']).max())
    np.testing.assert_allclose(efc[i], ref['f'][:26], rtol=1e-8, atol=1e-8 * (1 + np.abs(ref['f']).max()))
    for k in range(5):
      np.testing.assert_allclose(att[i, k], lm.attachment(ref['pos'], ref['quat'], k)[0], atol=1e-13)
    nlim += int(ref['active'][6:].sum())
  assert nlim > 20           # joint-limit rows were exercised


def test_steps_match_reference_and_track_the_mocap(env):
  torch, dm, lm = env
  n, nsub, iters = 8, 5, 30
  qpos, qvel, mp, mq, ctrl = random_states(n, 1)
  qvel *= 0
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
  dq, dv, dmp, dmq, dc = t(qpos), t(qvel), t(mp), t(mq), t(ctrl)
  att = torch.empty(n, 5, 3, dtype=torch.float64, device='cuda')
  rq, rv = qpos.copy(), qvel.copy()
  for it in range(iters):
    dm.step(dq, dv, dmp, dmq, dc, nsub=nsub, att_xpos=att)
    for i in range(n):
      for _ in range(nsub):
        rq[i], rv[i], out = lm.step(rq[i], rv[i], ctrl[i], mp[i], mq[i])
    if it in (0, 4, iters - 1):
      # trajectories are compared early (before chaotic growth of last-bit differences) and at the end loosely
      tol = 1e-8 if it < 5 else 1e-5
      np.testing.assert_allclose(dq.cpu().numpy(), rq, rtol=tol, atol=tol)
      np.testing.assert_allclose(dv.cpu().numpy(), rv, rtol=tol, atol=tol * 10)
  hand = att.cpu().numpy()[:, 0]
  assert np.abs(hand - mp).max() < 0.02          # the weld pulled the hand to the mocap target (2 cm: still settling)
  assert (dq.cpu().numpy()[:, 1] <= -0.5 + 2e-3).all()
