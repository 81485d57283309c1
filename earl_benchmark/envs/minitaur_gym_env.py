"""reference: earl_benchmark/envs/minitaur_gym_env.py (classes MinitaurBulletEnv :53, GoalConditionedMinitaurBulletEnv :451)"""
from earl_benchmark_amd.envs.minitaur import Minitaur

GoalConditionedMinitaurBulletEnv = Minitaur
MinitaurBulletEnv = Minitaur
