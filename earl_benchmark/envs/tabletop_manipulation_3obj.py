"""reference: earl_benchmark/envs/tabletop_manipulation_3obj.py (class TabletopManipulation :19)"""
from earl_benchmark_amd.envs.tabletop_3obj import TabletopManipulation  # noqa: F401
