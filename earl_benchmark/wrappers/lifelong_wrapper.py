"""reference: earl_benchmark/wrappers/lifelong_wrapper.py:8 (class LifelongWrapper)"""
from earl_benchmark_amd.wrappers import LifelongWrapper, Wrapper  # noqa: F401
