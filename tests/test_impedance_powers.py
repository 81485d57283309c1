"""The kitchen's and the minitaur's kernels evaluate MuJoCo's impedance with csrc/physics_math.h imp_p2 (solimp powers 1 and 2; reference: mujoco engine_core_constraint.c
getimpedance, restated in oracle/physics_oracle.c): the host side refuses tables with any other power for them, and every shipped table passes."""
import numpy as np
import pytest

from earl_benchmark_amd import _abi, physics


def test_other_powers_are_refused():
  ok = {'jeq_solimp': np.array([[0.9, 0.95, 0.001, 0.5, 2.0], [0.9, 0.95, 0.001, 0.5, 1.0], [0.9, 0.9, 0.001, 0.5, 3.0]])}      # (d0 == dwidth: the power is never used)
  physics.check_impedance_powers(ok, ('jeq_solimp', 'absent'), 'ok')
  bad = {'jnt_solimp': np.array([[0.9, 0.95, 0.001, 0.5, 2.0], [0.9, 0.95, 0.001, 0.5, 3.0]])}
  with pytest.raises(_abi.EarlHipError, match='row 1 has solimp power 3'):
    physics.check_impedance_powers(bad, ('jnt_solimp',), 'bad')


@pytest.mark.parametrize('name', ['kitchen', 'minitaur', 'sawyer_door', 'sawyer_peg'])
def test_shipped_tables_load(name):
  s, d = physics.load_link_model(name)
  physics.load_collision_model(d)
  if len(d['parent']) > 16:
    for k in ('jnt_solimp', 'weld_solimp', 'jeq_solimp', 'con_solimp', 'col_cls_solimp'):
      if k in d and len(d[k]):
        assert set(np.unique(np.asarray(d[k]).reshape(-1, 5)[:, 4])) <= {1.0, 2.0}, k
