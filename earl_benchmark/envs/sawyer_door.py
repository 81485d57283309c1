"""reference: earl_benchmark/envs/sawyer_door.py (class SawyerDoorV2 :18; initial_states / goal_states :13-16)"""
from earl_benchmark_amd import tables as _tables
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor

SawyerDoorV2 = SawyerDoor
initial_states = _tables.initial_states('sawyer_door')
goal_states = _tables.goal_states('sawyer_door')
