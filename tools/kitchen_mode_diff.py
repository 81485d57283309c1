"""Where two launch modes of the kitchen's fused rollout part ways, if they do: the first (env step, env, observation entry) whose bits differ between the packed launch (two envs
per wave) and each of the small-batch modes (include/earl_physics.h earl_debug_set_solo), on the workload of tests/test_kitchen_gpu.py::test_fused_rollout_equals_stepping_bit_for_bit
(203 envs, 200 env steps, forty hands pushed down onto the fixtures).  The tests assert equality; this prints WHERE it breaks -- an env without contacts from step 0 points at the
dynamics, an env at its first contact at the collision / contact rows (the four-waves-per-env form first came out 1e-13 off there).   python tools/kitchen_mode_diff.py [n] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd import _abi
from earl_benchmark_amd.envs.kitchen import Kitchen

nums = [int(x) for x in sys.argv[1:] if x.isdigit()]
n, T = (nums + [203, 200])[:2] if len(nums) < 2 else nums[:2]
lib = _abi.load()
g = torch.Generator(device='cuda').manual_seed(9)
acts = torch.rand(T, n, 9, generator=g, device='cuda') * 2 - 1
acts[:, :40, 2] -= 0.6
res = {}
try:
  for mode in (0, 1, 2, 3, 4):
    lib.earl_debug_set_solo(mode)
    env = Kitchen(num_envs=n, seed=21)
    env.reset()
    res[mode] = torch.nan_to_num(env.rollout(acts)['obs'].clone(), nan=123.0)
finally:
  lib.earl_debug_set_solo(-1)
for mode, name in ((1, 'one env per wave'), (2, 'one env per workgroup'), (3, 'four waves per env'), (4, 'two waves per env, two envs per workgroup')):
  bad = (res[mode].view(torch.int64) != res[0].view(torch.int64)).nonzero()
  if len(bad) == 0:
    print(f'{name}: the bits of the packed launch ({n} envs x {T} env steps)')
    continue
  t0 = int(bad[:, 0].min())
  first = bad[bad[:, 0] == t0]
  print(f'{name}: differs from the packed launch in {len(set(bad[:, 1].tolist()))} envs, largest difference {float((res[mode] - res[0]).abs().max()):.3g}; first at env step {t0}: '
        f'envs {sorted(set(first[:, 1].tolist()))[:10]}, observation entries {sorted(set(first[:, 2].tolist()))[:12]}')
