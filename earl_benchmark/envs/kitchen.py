"""reference: earl_benchmark/envs/kitchen.py (class Kitchen :86; goal_states :28; initial_states: dict by task :57-85)"""
from earl_benchmark_amd import tables as _tables
from earl_benchmark_amd.envs.kitchen import Kitchen  # noqa: F401

goal_states = _tables.goal_states('kitchen')
initial_states = {k: _tables.get('kitchen_task_' + k) for k in _tables.kitchen_tasks()}
