"""Does the start of the minitaur's active-set passes change anything but their number?  The shipped library (a contact slot that holds the same collision pair as at the
timestep before starts from the edge set its passes ended with: csrc/minitaur_stepper.h C3) against a build with the rule of rounds 2 - 5 (-DEARL_MT_NO_CARRY: the set the previous
solution predicts), same states, same actions: exact checksums (sums of the bit patterns as int64) of every output and of the final state, and the time of the launch.
   python tools/bench_mt_variant.py --build nocarry -DEARL_MT_NO_CARRY      (here)
   python tools/mt_start_rule_check.py [N] [T]                              (GPU box)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(tag, n, T):
  import torch
  from earl_benchmark_amd import _abi
  if tag != 'ship':
    _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_mt_{tag}.so')
  from earl_benchmark_amd.envs.minitaur import Minitaur
  lib = _abi.load()
  out = {}
  for duo in (0, 1):
    lib.earl_debug_set_minitaur_duo(duo)
    env = Minitaur(num_envs=n, seed=1234, scalar_api=False)
    acts = (torch.rand(T, n, 8, generator=torch.Generator(device='cuda').manual_seed(99), device='cuda') * 2 - 1).float()
    env.reset(); r = env.rollout(acts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.reset(); r = env.rollout(acts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bits = lambda x: int(x.contiguous().view(torch.int64).sum()) if x.dtype == torch.float64 else int(x.to(torch.int64).sum())
    out['two waves per SIMD' if duo else 'one wave per SIMD'] = {
      'ms': dt * 1e3, 'failed_env_steps': int(env.fail_count.sum()),
      'bit_sums': {**{k: bits(v) for k, v in r.items() if hasattr(v, 'dtype')}, 'qpos': bits(env.qpos), 'qvel': bits(env.qvel), 'observed_torque': bits(env.observed_torque)}}
  import numpy as np
  np.save(os.path.join(ROOT, 'gpurun_out', f'_start_rule_{tag}.npy'), r['obs'].view(torch.int64).sum(-1).cpu().numpy())      # [T, n]: one exact word per env step
  lib.earl_debug_set_minitaur_duo(-1)
  print('RESULT ' + json.dumps(out), flush=True)


def main():
  if len(sys.argv) > 1 and sys.argv[1] == '--child':
    return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
  nums = [int(x) for x in sys.argv[1:] if x.isdigit()]
  n, T = (nums + [4096, 1000])[:2] if len(nums) < 2 else nums[:2]
  res = {}
  for tag in ('ship', 'nocarry'):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tag, str(n), str(T)], capture_output=True, text=True, timeout=1200)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
    res[tag] = json.loads(lines[-1][7:]) if lines else {'error': r.stderr[-600:]}
  same = all('error' not in res[t] for t in res) and all(res['ship'][k]['bit_sums'] == res['nocarry'][k]['bit_sums'] for k in res['ship']) \
      and res['ship']['one wave per SIMD']['bit_sums'] == res['ship']['two waves per SIMD']['bit_sums']
  import numpy as np
  fa, fb = (os.path.join(ROOT, 'gpurun_out', f'_start_rule_{t}.npy') for t in ('ship', 'nocarry'))
  where = None
  if os.path.exists(fa) and os.path.exists(fb):
    dif = np.load(fa) != np.load(fb)                       # [T, n]
    envs = dif.any(0)
    first = np.where(envs, dif.argmax(0), T)
    where = {'envs_that_differ_at_some_step': int(envs.sum()), 'first_differing_step_of_those': sorted(int(x) for x in first[envs])[:40],
             'env_ids': [int(x) for x in np.nonzero(envs)[0][:40]], 'env_steps_before_the_first_difference': int(np.minimum(first, T).sum()), 'env_steps': int(n * T)}
    os.remove(fa); os.remove(fb)
  print(json.dumps({'envs': n, 'env_steps': T, 'all_bit_sums_equal': same, 'where': where, **res}, indent=1))


if __name__ == '__main__':
  main()
