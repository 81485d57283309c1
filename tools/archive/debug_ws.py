import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import hip_harness as hx
from earl_benchmark_amd import _abi
from oracle import tabletop_oracle as orc
lib = _abi.load()
for n, T in [(64, 5), (64, 8), (64, 12), (64, 16), (16, 9)]:
  rng = np.random.default_rng(0)
  acts = rng.uniform(-1, 1, size=(T, n, 3)).astype(np.float32)
  outs = {}
  for impl in (0, 1):
    h = hx.HipTabletop(n, horizon=1000, seed=1)
    h.reset()
    lib.earl_debug_set_rollout_impl(impl)
    outs[impl] = h.rollout(acts)
    lib.earl_debug_set_rollout_impl(0)
  names = ['obs', 'rew', 'done', 'succ']
  for nm, x, y in zip(names, outs[0], outs[1]):
    bad = (x != y)
    if bad.any():
      idx = np.argwhere(bad)
      print(n, T, nm, 'mismatches', bad.sum(), 'steps', sorted(set(idx[:, 0]))[:20], 'envs', sorted(set(idx[:, 1]))[:10], 'cols', sorted(set(idx[:, 2])) if idx.shape[1] > 2 else '')
      t, e = idx[0][0], idx[0][1]
      print('   first: t', t, 'env', e, 'ws', x[t, e], 'plain', y[t, e])
    else:
      print(n, T, nm, 'ok')
