"""Loader surface that needs no GPU: configs, initial/goal-state tables and demonstrations (layout of the reference)."""
import numpy as np
import pytest

import earl_benchmark_amd as eb
from conftest import load_golden


def bare(name, **kw):
  # the real constructor: envs are built on the first get_envs(), so tables / demos need no GPU (and exist for every env name)
  return eb.EARLEnvs(name, reward_type='dense' if name == 'kitchen' else 'sparse', **kw)


def test_constructor_needs_no_gpu_and_keeps_the_reference_errors():
  for name in eb.deployment_eval_config:
    L = bare(name)
    assert L._env_name == name and L._envs is None
    bare(name, setup_as_lifelong_learning=True)
  with pytest.raises(ValueError, match='only supports dense'):
    eb.EARLEnvs('kitchen', reward_type='sparse')          # reference: envs/kitchen.py:91-92
  with pytest.raises(KeyError):
    eb.EARLEnvs('no_such_env')


def test_configs_match_reference_values():
  d, c = eb.deployment_eval_config, eb.continuing_eval_config
  assert d['tabletop_manipulation'] == {'num_initial_state_samples': 1, 'num_goals': 4, 'train_horizon': 200000, 'eval_horizon': 200}
  assert (d['sawyer_door']['eval_horizon'], d['sawyer_peg']['eval_horizon'], d['kitchen']['eval_horizon'], d['minitaur']['eval_horizon']) == (300, 200, 400, 1000)
  assert [c[k]['goal_change_frequency'] for k in ('tabletop_manipulation', 'sawyer_door', 'sawyer_peg', 'kitchen', 'minitaur')] == [400, 600, 400, 800, 2000]
  assert d['sawyer_peg']['num_initial_state_samples'] == 15 and d['kitchen']['task'] == 'all_pairs'


@pytest.mark.parametrize('name,ishape,gshape', [('tabletop_manipulation', (1, 6), (4, 6)), ('sawyer_door', (1, 7), (1, 7)),
                                                ('sawyer_peg', (15, 7), (1, 7)), ('kitchen', (6, 23), (1, 23))])
def test_tables(name, ishape, gshape):
  g = load_golden('loader_tables')
  L = bare(name)
  np.testing.assert_array_equal(L.get_initial_states(), g[f'{name}_initial_states'])
  np.testing.assert_array_equal(L.get_goal_states(), g[f'{name}_goal_states'])
  assert L.get_initial_states().shape == ishape and L.get_goal_states().shape == gshape


def test_tabletop_tables_match_env_module_constants():
  from earl_benchmark_amd.envs import tabletop
  g = load_golden('loader_tables')
  np.testing.assert_array_equal(tabletop.initial_states, g['tabletop_manipulation_initial_states'])
  np.testing.assert_array_equal(tabletop.goal_states, g['tabletop_manipulation_goal_states'])
  # task order r,k,g,b -> goal_states rows 0,3,1,2
  rows = tabletop.task_goal_rows('rc_r-rc_k-rc_g-rc_b')
  np.testing.assert_array_equal(rows, tabletop.goal_states[[0, 3, 1, 2]])
  assert bare('minitaur').get_goal_states() is None


@pytest.mark.parametrize('name,n_f,n_r,d,a', [('tabletop_manipulation', 1278, 1256, 12, 3), ('sawyer_door', 395, 700, 14, 4),
                                              ('sawyer_peg', 683, 1132, 14, 4)])
def test_demonstrations_layout(name, n_f, n_r, d, a):
  L = bare(name)
  assert L.has_demos()
  fwd, rev = L.get_demonstrations()
  for demo, n in ((fwd, n_f), (rev, n_r)):
    assert set(demo) == {'observations', 'actions', 'rewards', 'terminals', 'next_observations', 'infos'}
    assert demo['observations'].shape == (n, d) and demo['observations'].dtype == np.float32
    assert demo['actions'].shape == (n, a) and demo['rewards'].shape == (n, 1) and demo['terminals'].dtype == bool
    assert demo['next_observations'].shape == (n, d)
  assert not bare('kitchen').has_demos() and bare('kitchen').get_demonstrations() is None


def test_sawyer_sparse_rule_holds_on_the_demos():
  """sparse reward == (||object - goal|| <= radius) on every recorded row."""
  for name, radius in (('sawyer_door', 0.02), ('sawyer_peg', 0.05)):
    for demo in bare(name).get_demonstrations():
      o = demo['next_observations']
      assert ((np.linalg.norm(o[:, 4:7] - o[:, 11:14], axis=1) <= radius) == (demo['rewards'][:, 0] == 1)).all()


def test_reference_import_name_works_unchanged():
  """`import earl_benchmark` (reference README.md:21-31) resolves to this build: same loader, same tables, no GPU needed for them"""
  import earl_benchmark
  import earl_benchmark_amd
  assert earl_benchmark.EARLEnvs is earl_benchmark_amd.EARLEnvs
  assert earl_benchmark.deployment_eval_config is earl_benchmark_amd.deployment_eval_config
  loader = earl_benchmark.EARLEnvs('tabletop_manipulation', reward_type='sparse')
  assert loader.get_initial_states().shape == (1, 6) and loader.get_goal_states().shape == (4, 6)
  fwd, rev = loader.get_demonstrations()
  assert set(fwd) == set(rev) == {'observations', 'actions', 'rewards', 'terminals', 'next_observations', 'infos'}
  assert earl_benchmark.wrappers.PersistentStateWrapper is earl_benchmark_amd.wrappers.PersistentStateWrapper
  assert earl_benchmark.tables is earl_benchmark_amd.tables


def test_the_references_literal_import_statements_resolve():
  """reference earl_benchmark/__init__.py:7-8 and :114-136 import sub-MODULES by name; they are real packages of the shim (a module-level
  __getattr__ is not consulted for `import a.b` / `from a.b import c`)"""
  pytest = __import__('pytest')
  pytest.importorskip('torch')
  from earl_benchmark.wrappers import persistent_state_wrapper
  from earl_benchmark.wrappers import lifelong_wrapper
  from earl_benchmark.envs import tabletop_manipulation
  from earl_benchmark.envs import sawyer_door
  from earl_benchmark.envs import sawyer_peg
  from earl_benchmark.envs import kitchen
  from earl_benchmark.envs import minitaur_gym_env
  import earl_benchmark.wrappers
  import earl_benchmark_amd
  assert persistent_state_wrapper.PersistentStateWrapper is earl_benchmark_amd.wrappers.PersistentStateWrapper
  assert lifelong_wrapper.LifelongWrapper is earl_benchmark_amd.wrappers.LifelongWrapper
  assert earl_benchmark.wrappers.PersistentStateWrapper is earl_benchmark_amd.wrappers.PersistentStateWrapper
  assert tabletop_manipulation.TabletopManipulation.__module__ == 'earl_benchmark_amd.envs.tabletop'
  assert sawyer_door.SawyerDoorV2.__name__ == 'SawyerDoor' and sawyer_peg.SawyerPegV2.__name__ == 'SawyerPeg'
  assert kitchen.Kitchen.__module__ == 'earl_benchmark_amd.envs.kitchen' and minitaur_gym_env.GoalConditionedMinitaurBulletEnv.__name__ == 'Minitaur'
  # the module-level tables the reference's loader reads (`sawyer_door.initial_states`, ...: reference earl_benchmark/__init__.py:194-237)
  import numpy as np
  for mod, name in ((tabletop_manipulation, 'tabletop_manipulation'), (sawyer_door, 'sawyer_door'), (sawyer_peg, 'sawyer_peg')):
    loader = earl_benchmark_amd.EARLEnvs(name, reward_type='sparse')
    np.testing.assert_array_equal(mod.initial_states, loader.get_initial_states())
    np.testing.assert_array_equal(mod.goal_states, loader.get_goal_states())
  np.testing.assert_array_equal(kitchen.goal_states, earl_benchmark_amd.EARLEnvs('kitchen', reward_type='dense').get_goal_states())
  assert set(kitchen.initial_states) >= {'microwave', 'light_switch', 'slide_cabinet', 'hinge_cabinet', 'all_pairs'}
  # the 3-object variant's module has the same two tables (reference envs/tabletop_manipulation_3obj.py:11-17; values checked against the reference's by
  # tests/golden/make_golden.py: the first goal row / the reset state of the recorded rollouts)
  from earl_benchmark.envs import tabletop_manipulation_3obj as t3
  assert t3.initial_states.shape == (1, 10) and t3.goal_states.shape == (1, 10) and t3.TabletopManipulation.NOBJ == 3
  np.testing.assert_array_equal(t3.initial_states[0], [0.0, 0.0, 2.5, 0.0, 2.5, -1.0, 2.5, 1.0, -1., -1.])
  np.testing.assert_array_equal(t3.goal_states[0], [0.0, 0.0, 0.0, -2.0, 0.0, 2.0, -2.5, 1.0, -1., -1.])
