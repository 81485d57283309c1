"""`import earl_benchmark` -- the reference's import name (reference README.md:21-31, earl_benchmark/__init__.py:83-110) served by this
build: everything is earl_benchmark_amd (the MI355X-native hot path behind the same loader / gym-style API); this package only re-exports
it, so code written against the reference runs unchanged:

    import earl_benchmark
    env_loader = earl_benchmark.EARLEnvs('tabletop_manipulation', reward_type='sparse')
    train_env, eval_env = env_loader.get_envs()

No environment arithmetic lives here."""
import sys as _sys

import earl_benchmark_amd as _impl
from earl_benchmark_amd import *  # noqa: F401,F403
from earl_benchmark_amd import (EARLEnvs, UnpinnedDynamicsWarning, continuing_eval_config, deployment_eval_config, load_demo,  # noqa: F401
                                __version__)

# The reference's sub-packages are REAL packages here (earl_benchmark/envs/, earl_benchmark/wrappers/): `from earl_benchmark.envs import tabletop_manipulation`,
# `from earl_benchmark.wrappers import persistent_state_wrapper` (reference earl_benchmark/__init__.py:7-8, :114-136) work as written.  (A module-level
# __getattr__ is not consulted by the import system for submodules: ADVICE r03.)  They import torch, so they are resolved on first use, not here: tables
# and demonstrations work on machines without a GPU stack.
_ALIASES = {'tables': 'earl_benchmark_amd.tables', 'sharding': 'earl_benchmark_amd.sharding', 'glue': 'earl_benchmark_amd.glue'}


def __getattr__(name):
  import importlib
  if name in ('envs', 'wrappers'):
    return importlib.import_module(__name__ + '.' + name)
  if name in _ALIASES:
    mod = importlib.import_module(_ALIASES[name])
    _sys.modules[__name__ + '.' + name] = mod
    return mod
  return getattr(_impl, name)
