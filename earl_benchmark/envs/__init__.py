"""earl_benchmark.envs -- the reference's module names (earl_benchmark/envs/*.py), each a re-export of this build's env class and tables."""
