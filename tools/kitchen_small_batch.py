"""how long does a SHARD of the kitchen / minitaur strong-scaling batches take on one GPU?  reset + one fused launch of the full horizon for
n = the 8-, 4-, 2- and 1-GPU shard sizes (same seeds per env id: the shard's envs are the batch's first n).   python tools/kitchen_small_batch.py > profiles/r05_kitchen_small_batch.txt
Round 5: every shard size under each small-batch mode of the launch (include/earl_physics.h earl_debug_set_solo: 0 = two envs per wave, 1 = one env per wave, 2 = one env per
workgroup, -1 = what the launcher picks), and a check that the modes' outputs are bit-identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from earl_benchmark_amd.envs.kitchen import Kitchen
from earl_benchmark_amd.envs.minitaur import Minitaur
from earl_benchmark_amd.wrappers import PersistentStateWrapper


from earl_benchmark_amd import _abi
LIB = _abi.load()


def run(make, n, T, adim, reps=2, mode=-1, keep=None):
  LIB.earl_debug_set_solo(mode); LIB.earl_debug_set_solo_mt(mode)
  env = PersistentStateWrapper(make(n), T)
  g = torch.Generator(device='cuda').manual_seed(77)
  acts = (torch.rand(T, n, adim, generator=g, device='cuda') * 2 - 1).float()
  env.reset(); out = env.unwrapped.rollout(acts)
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(reps):
    t0 = time.perf_counter()
    env.reset(); env.unwrapped.rollout(acts, out=out)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
  LIB.earl_debug_set_solo(-1); LIB.earl_debug_set_solo_mt(-1)
  if keep is not None:
    keep.append({k: v.clone() for k, v in out.items() if torch.is_tensor(v)})
  return best


print(f'device: {torch.cuda.get_device_name(0)}')
for name, make, N, T, adim in (('kitchen', lambda n: Kitchen(num_envs=n, seed=1234), 2048, 400, 9),
                               ('minitaur', lambda n: Minitaur(num_envs=n, seed=1234, scalar_api=False), 4096, 1000, 8)):
  full = None
  for w in (1, 2, 4, 8):
    n = N // w
    keep = []
    times = {mode: run(make, n, T, adim, mode=mode, keep=keep) for mode in ((0, 1, 2, 3, 4, -1) if name == 'kitchen' and n <= 256 else ((0, 1, 2, 4, -1) if name == 'kitchen' and n <= 512 else (0, 1, 2, -1)))}
    same = all(all(torch.equal(keep[0][k], o[k]) for k in keep[0]) for o in keep[1:])
    dt = times[-1]
    full = dt if full is None else full
    duo = (f', four waves per env {times[3] * 1e3:.1f} ms' if 3 in times else '') + (f', two waves per env (two envs per workgroup) {times[4] * 1e3:.1f} ms' if 4 in times else '')
    print(f'{name}: shard of {w} GPU(s) = {n:5d} envs x {T} steps: {dt * 1e3:8.1f} ms per reset + launch = {dt / full:5.2f} x the {N}-env launch; '
          f'{w} such GPUs would deliver {N * T / dt / 1e6:6.2f} M env-steps/s ({full / dt:4.2f} x one GPU) | two envs per wave {times[0] * 1e3:.1f} ms, one env per wave '
          f'{times[1] * 1e3:.1f} ms, one env per workgroup {times[2] * 1e3:.1f} ms{duo}; outputs bit-identical across the modes: {same}', flush=True)
