"""Env factory handed to `bench.py --test-env-factory` by tests/test_bench_launch.py: the ValidatorEnv of tests/test_bench_sequence.py (the HIP library's own
argument validator + synthetic outputs on host tensors), so that bench.py's whole launch path -- main() -> self_launch -> torch.distributed.run -> main() on every
rank -> time_rollouts -> the JSON line -- runs here without a GPU, over gloo.  Test infrastructure only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make(n, T, rank, world):
  del world
  from test_bench_sequence import ValidatorEnv
  return ValidatorEnv(n, T, rank * n, delay=0.0)
