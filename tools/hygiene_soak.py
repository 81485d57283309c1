"""Bench-size version of tests/test_lds_hygiene_gpu.py (one-off soak): every env at its BASELINE size and horizon, LDS pre-filled with NaNs vs zeros before the
constructor, the reset and the fused rollout -- outputs must be bit-identical and the failure guard silent."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib = C.CDLL(os.path.join(ROOT, 'tests', 'liblds_poison.so'))
lib.lds_poison.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(8, dtype=torch.int64, device='cuda')


def poison(p):
  assert lib.lds_poison(p, 2048, sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
  torch.cuda.synchronize()


from earl_benchmark_amd.envs.minitaur import Minitaur
from earl_benchmark_amd.envs.kitchen import Kitchen
from earl_benchmark_amd.envs.sawyer_door import SawyerDoor
from earl_benchmark_amd.envs.sawyer_peg import SawyerPeg
cases = [('minitaur 4096x1000', lambda: Minitaur(num_envs=4096, scalar_api=False, seed=9), 1000, 8, 250),
         ('kitchen 2048x400', lambda: Kitchen(num_envs=2048, scalar_api=False, seed=9), 400, 9, 100),
         ('sawyer_door 8192x300', lambda: SawyerDoor(num_envs=8192, scalar_api=False, seed=9), 300, 4, 100),
         ('sawyer_door (at goal) 8192x300', lambda: SawyerDoor(num_envs=8192, scalar_api=False, seed=9, reset_at_goal=True), 300, 4, 100),
         ('sawyer_peg 8192x200', lambda: SawyerPeg(num_envs=8192, scalar_api=False, seed=9), 200, 4, 100),
         ('sawyer_peg (at goal) 8192x200', lambda: SawyerPeg(num_envs=8192, scalar_api=False, seed=9, reset_at_goal=True), 200, 4, 100)]
for name, make, T, adim, chunk in cases:
  res = {}
  for tag, pat in (('zero', 0), ('nan', 0x7FF8000000000000)):
    poison(pat); env = make(); poison(pat); env.reset()
    g = torch.Generator(device='cuda').manual_seed(4)
    digest, t0 = [], time.time()
    for k in range(T // chunk):
      acts = torch.rand(chunk, env.num_envs, adim, generator=g, device='cuda') * 2 - 1
      poison(pat)
      out = env.rollout(acts)
      digest.append(out['obs'].clone())
    res[tag] = (digest, int(env.fail_count.sum()), bool(all(torch.isfinite(d).all() for d in digest)))
    del env
  same = all(torch.equal(a, b) for a, b in zip(res['zero'][0], res['nan'][0]))
  print(f'{name:32s} identical: {same}; rolled-back steps zero/nan: {res["zero"][1]}/{res["nan"][1]}; finite: {res["zero"][2]}/{res["nan"][2]}', flush=True)
