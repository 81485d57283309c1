"""What does floating-point contraction buy the four articulated-body steppers?  The shipped units compile under `#pragma clang fp contract(fast)`
(csrc/physics.hip); tools/build_phys_variant.sh nocontract -DEARL_PHYS_NO_CONTRACT builds them under -ffp-contract=off (explicit fma() only, as the tabletop path and the
C restatement are built).  This script times the bench's four launches through both libraries on one MI355X and reports how far the observations of the two builds
are apart after one full-horizon rollout.

usage (GPU box): python tools/contraction_cost.py [tag ...]        (default tags: ship nocontract)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {'sawyer_door': (8192, 300, 4), 'sawyer_peg': (8192, 200, 4), 'kitchen': (2048, 400, 9), 'minitaur': (4096, 1000, 8)}


def child(tag, outdir):
  import torch
  from earl_benchmark_amd import _abi
  if tag != 'ship':
    _abi.LIB_PATH = os.path.join(ROOT, 'tools', 'ubench', f'libearl_phys_{tag}.so')
  from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
  from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
  from earl_benchmark_amd.envs.kitchen import Kitchen
  from earl_benchmark_amd.envs.minitaur import Minitaur
  res = {}
  for name, (n, T, A) in SHAPES.items():
    mk = {'sawyer_door': lambda: SawyerDoor(num_envs=n, seed=1234), 'sawyer_peg': lambda: SawyerPeg(num_envs=n, seed=1234),
          'kitchen': lambda: Kitchen(num_envs=n, seed=1234), 'minitaur': lambda: Minitaur(num_envs=n, seed=1234, scalar_api=False)}[name]
    env = mk()
    g = torch.Generator(device='cuda').manual_seed(99)
    acts = (torch.rand(T, n, A, generator=g, device='cuda') * 2 - 1).float()
    env.reset(); r = env.rollout(acts)
    torch.cuda.synchronize()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
      env.reset(); r = env.rollout(acts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    obs = r['obs'] if isinstance(r, dict) else r[0]
    torch.save(obs[::max(1, T // 20)].double().cpu(), os.path.join(outdir, f'{name}_{tag}.pt'))      # every T/20-th row: enough to see where the builds part
    res[name] = {'ms': dt * 1e3, 'env_steps_per_s': n * T / dt, 'obs_checksum': float(obs.double().sum())}
    del env, acts, r, obs
    torch.cuda.empty_cache()
  print('RESULT ' + json.dumps(res), flush=True)


def main():
  if len(sys.argv) > 2 and sys.argv[1] == '--child':
    return child(sys.argv[2], sys.argv[3])
  import tempfile
  tags = sys.argv[1:] or ['ship', 'nocontract']
  out = {}
  with tempfile.TemporaryDirectory() as d:
    for tag in tags:
      r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tag, d], capture_output=True, text=True, timeout=1500)
      lines = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
      out[tag] = json.loads(lines[-1][7:]) if lines else {'error': r.stderr[-800:]}
    import torch
    for tag in tags[1:]:
      for name in SHAPES:
        a, b = os.path.join(d, f'{name}_{tags[0]}.pt'), os.path.join(d, f'{name}_{tag}.pt')
        if os.path.exists(a) and os.path.exists(b):
          x, y = torch.load(a), torch.load(b)
          dif = (x - y).abs().reshape(x.shape[0], -1).max(1).values
          out[tag][name]['max_abs_obs_diff_vs_' + tags[0]] = {'first_sampled_row': float(dif[0]), 'last_sampled_row': float(dif[-1]), 'max': float(dif.max())}
          out[tag][name]['time_ratio_vs_' + tags[0]] = out[tag][name]['ms'] / out[tags[0]][name]['ms']
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
