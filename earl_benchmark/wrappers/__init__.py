"""earl_benchmark.wrappers -- the reference's module names (earl_benchmark/wrappers/*.py); also the classes themselves, as in earlier rounds."""
from earl_benchmark_amd.wrappers import LifelongWrapper, PersistentStateWrapper, Wrapper  # noqa: F401
