"""reference: earl_benchmark/wrappers/persistent_state_wrapper.py:8 (class PersistentStateWrapper)"""
from earl_benchmark_amd.wrappers import PersistentStateWrapper, Wrapper  # noqa: F401
