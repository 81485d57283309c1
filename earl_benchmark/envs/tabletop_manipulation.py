"""reference: earl_benchmark/envs/tabletop_manipulation.py (class TabletopManipulation :18; initial_states / goal_states :11-16)"""
from earl_benchmark_amd import tables as _tables
from earl_benchmark_amd.envs.tabletop import TabletopManipulation  # noqa: F401

initial_states = _tables.initial_states('tabletop_manipulation')
goal_states = _tables.goal_states('tabletop_manipulation')
