"""reference: earl_benchmark/envs/tabletop_manipulation_3obj.py (class TabletopManipulation :19; initial_states / goal_states :11-17)"""
from earl_benchmark_amd.envs.tabletop_3obj import TabletopManipulation, goal_states, initial_states  # noqa: F401
