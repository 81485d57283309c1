"""reference: earl_benchmark/envs/sawyer_peg.py (class SawyerPegV2 :60; initial_states / goal_states :18-58)"""
from earl_benchmark_amd import tables as _tables
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg

SawyerPegV2 = SawyerPeg
initial_states = _tables.initial_states('sawyer_peg')
goal_states = _tables.goal_states('sawyer_peg')
